#!/usr/bin/env python
"""profiles/pmc_<workload>.json from the separate rocprofv3 PMC passes of `python bench.py --workload
<w> --no-extra --no-cpu-baseline` (scripts/pmc_pass.sh: FETCH_SIZE, WRITE_SIZE and SQ passes):

    python scripts/pmc_to_json.py <workload> <fetch summary.json> <write summary.json> <sq summary.json>

Per kernel of the fused step: HBM bytes per launch with the gfx950 correction of
MI355X_MICROARCH.md (FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads ->
doubled; WRITE_SIZE as reported; both in KB), VALU / SALU wave-instructions per launch."""
import hashlib
import json
import os
import sys


def csrc_sha256(root=None):
    """Fingerprint of the kernel sources the counters were taken from (bench.py refuses counters whose
    fingerprint differs from the tree's: a rebuilt kernel must not carry stale traffic figures)."""
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "pytorchltr_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".inc")):
            with open(os.path.join(d, name), "rb") as fh:
                h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()

KEYS = {"linear_regtile_kernel": "linear_regtile", "linear_cluster_kernel": "linear_cluster_kernel",
        "linear_parts_kernel": "linear_parts_kernel", "stream_probe_kernel": "stream_probe_kernel",
        "linear_pairwise_kernel": "linear_pairwise_kernel", "pairwise_loss_kernel": "pairwise_loss_kernel",
        "linear_reduce_kernel": "linear_reduce_kernel", "metric_kernel": "metric_kernel"}


def pick(doc, needle):
    best = None
    for name, rec in doc.items():
        if needle in name and (best is None or rec.get("calls", 0) > best[1].get("calls", 0)):
            best = (name, rec)
    return best


def main(workload, fetch, write, sq):
    docs = [json.load(open(p)) for p in (fetch, write, sq)]
    out = {"_source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | SQ_* (separate passes) -- "
                      "python bench.py --workload %s --no-extra --no-cpu-baseline; FETCH_SIZE doubled "
                      "(gfx950: wide coalesced reads are tallied at half their bytes, MI355X_MICROARCH.md)" % workload,
           "_csrc_sha256": csrc_sha256()}
    for key, needle in KEYS.items():
        f, w, s = (pick(d, needle) for d in docs)
        if not f and not s:
            continue
        rec = {"kernel": (f or s)[0][:120]}
        if f and "counters" in f[1] and "FETCH_SIZE" in f[1]["counters"]:
            rec["FETCH_SIZE_KB_raw"] = f[1]["counters"]["FETCH_SIZE"]
            rec["hbm_read_bytes_per_launch"] = 2.0 * 1024.0 * f[1]["counters"]["FETCH_SIZE"]
        if w and "counters" in w[1] and "WRITE_SIZE" in w[1]["counters"]:
            rec["WRITE_SIZE_KB_raw"] = w[1]["counters"]["WRITE_SIZE"]
            rec["hbm_write_bytes_per_launch"] = 1024.0 * w[1]["counters"]["WRITE_SIZE"]
        if "hbm_read_bytes_per_launch" in rec and "hbm_write_bytes_per_launch" in rec:
            rec["hbm_bytes_per_launch"] = rec["hbm_read_bytes_per_launch"] + rec["hbm_write_bytes_per_launch"]
        if s and "counters" in s[1]:
            c = s[1]["counters"]
            rec["valu_insts_per_launch"] = c.get("SQ_INSTS_VALU")
            rec["salu_insts_per_launch"] = c.get("SQ_INSTS_SALU")
            rec["lds_insts_per_launch"] = c.get("SQ_INSTS_LDS")
            rec["waves_per_launch"] = c.get("SQ_WAVES")
            rec["avg_us_under_pmc"] = s[1].get("avg_us")
        out[key] = rec
    with open("profiles/pmc_%s.json" % workload, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
