# gpurun_out/r06f (scripts/dev/r06_final.sh) -> profiles/: the files the docs cite
set -e
cd "$(dirname "$0")/../.."
S=gpurun_out/r06f; P=$S/prof
grep '^{' $S/bench.json | tail -1 > profiles/r06_bench_default.json
cp $P/trace_c2/summary.txt profiles/r06_c2_trace.txt
for w in c2 c3 c4 c5; do
  for k in sq fetch write; do cp $P/${w}_$k/summary.txt profiles/r06_${w}_$k.txt; done
  cp $P/pmc_$w.json profiles/pmc_$w.json
done
( echo "== rocprofv3 --kernel-trace --stats -- python scripts/lazy_trace.py lazy 3000"; cat $P/lazy/summary.txt
  echo "== ... eager 3000"; cat $P/eager/summary.txt ) > profiles/r06_lazy_trace.txt
( echo "== rocprofv3 --kernel-trace --stats -- python scripts/dropin_loop.py lazy 1000   (pytorchltr_amd.optim.SGD)"; cat $P/dropin_lazy/summary.txt
  echo "== ... torch 1000   (torch.optim.SGD, same model)"; cat $P/dropin_torch/summary.txt ) > profiles/r06_dropin_loop_trace.txt
cp $S/regtile_phases.txt profiles/r06_regtile_phases.txt
cp $S/cu_bytes.txt profiles/r06_cu_bytes.txt
cp $S/cluster_phases.txt profiles/r06_cluster_phases.txt
cp $S/sweep_b.jsonl profiles/r06_sweep_b.jsonl
cp $S/dp_stress.jsonl profiles/r06_dp_stress.jsonl
git status --short profiles | head -40
