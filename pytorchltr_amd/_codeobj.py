"""Kernel resource records of the built library (no GPU needed).

``kernel_records(path)`` reads the gfx950 code objects out of a shared library built by
``pytorchltr_amd.build`` -- the ``.hip_fatbin`` section holds one clang offload bundle per translation
unit -- and returns the per-kernel metadata the assembler left in their ELF notes (``.vgpr_count``,
``.vgpr_spill_count``, ``.sgpr_count``, ``.private_segment_fixed_size``, ``.group_segment_fixed_size``).
Used by tests/test_codeobj.py (no product kernel may spill) and scripts/dev.
"""
import os
import re
import struct
import subprocess
import tempfile

_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
_LLVM_BIN = os.environ.get("LTR_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def _tool(name):
    path = os.path.join(_LLVM_BIN, name)
    if not os.path.exists(path):
        raise FileNotFoundError("%s not found (set LTR_LLVM_BIN)" % path)
    return path


def _fatbin_bytes(lib_path):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "fatbin.bin")
        subprocess.check_call([_tool("llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib_path, out])
        with open(out, "rb") as fh:
            return fh.read()


def code_objects(lib_path, arch="gfx950"):
    """The device ELF images for `arch`, one per translation unit."""
    blob = _fatbin_bytes(lib_path)
    images = []
    pos = blob.find(_MAGIC)
    while pos >= 0:
        (count,) = struct.unpack_from("<Q", blob, pos + len(_MAGIC))
        cur = pos + len(_MAGIC) + 8
        for _ in range(count):
            offset, size, tlen = struct.unpack_from("<QQQ", blob, cur)
            triple = blob[cur + 24:cur + 24 + tlen].decode()
            cur += 24 + tlen
            if arch in triple and size > 0:
                images.append(blob[pos + offset:pos + offset + size])
        pos = blob.find(_MAGIC, pos + len(_MAGIC))
    return images


_KEYS = ("name", "vgpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count",
         "private_segment_fixed_size", "group_segment_fixed_size", "agpr_count", "max_flat_workgroup_size")


def kernel_records(lib_path, arch="gfx950"):
    """[{name, vgpr_count, vgpr_spill_count, ...}] for every kernel of the library (demangled names)."""
    records = []
    with tempfile.TemporaryDirectory() as tmp:
        for i, image in enumerate(code_objects(lib_path, arch)):
            path = os.path.join(tmp, "co%d.elf" % i)
            with open(path, "wb") as fh:
                fh.write(image)
            text = subprocess.run([_tool("llvm-readelf"), "--notes", path], check=True,
                                  stdout=subprocess.PIPE).stdout.decode()
            cur = None
            for line in text.splitlines():
                m = re.match(r"\s*(?:- )?\.(\w+):\s*(.*)$", line)
                if not m:
                    continue
                key, val = m.group(1), m.group(2).strip().strip("'")
                starts = line.lstrip().startswith("- ")
                if starts and (key == "agpr_count" or cur is None or key in cur):
                    # a new list element of amdhsa.kernels starts with '- .<first key>'
                    if cur is not None and "name" in cur:
                        records.append(cur)
                    cur = {}
                if cur is None:
                    continue
                if key in _KEYS and key not in cur:
                    cur[key] = val if key == "name" else int(val)
            if cur is not None and "name" in cur:
                records.append(cur)
    names = [r["name"] for r in records]
    if names:
        import shutil
        filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
        dem = names
        if filt:
            dem = subprocess.run([filt], input="\n".join(names).encode(), check=True,
                                 stdout=subprocess.PIPE).stdout.decode().splitlines()
        for r, d in zip(records, dem):
            r["demangled"] = d
    return records


if __name__ == "__main__":
    import sys
    from .build import LIB_PATH
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    for rec in kernel_records(LIB_PATH):
        if pat in rec.get("demangled", rec["name"]):
            print("%-90s vgpr %3d spill %3d sgpr %3d scratch %4d lds %6d" % (
                rec.get("demangled", rec["name"])[:90], rec.get("vgpr_count", -1), rec.get("vgpr_spill_count", -1),
                rec.get("sgpr_count", -1), rec.get("private_segment_fixed_size", -1),
                rec.get("group_segment_fixed_size", -1)))
