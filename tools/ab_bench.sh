#!/bin/bash
# A/B of engine-library variants on ONE box: tools/ab_bench.sh <out_dir> <variant.so> [<variant.so> ...]
# Runs bench.py --no-dropin on the in-tree library and on every variant (selected with AMP_ENGINE_LIB), interleaved, twice.  Variants: tools/build_variant.sh <name> <source.hip> -D...  ->  tools/bin/libamp_<name>.so.
set -u
out=$1; shift
mkdir -p "$out"
# variants are selected through AMP_ENGINE_LIB (humanoid_amp_amd/_native.py): the in-tree library is never touched
for rep in 1 2; do
  for v in base "$@"; do
    if [ "$v" = base ]; then lib=""; n=base; else lib=$(readlink -f "$v"); n=$(basename "$v" .so); fi
    AMP_ENGINE_LIB="$lib" timeout -k 10 200 python bench.py --no-dropin --no-configs --no-update --no-cpu-baseline --no-fp32-engine --steps 100 > "$out/bench_${n}_$rep.json" 2> "$out/bench_${n}_$rep.err"
    python - "$out/bench_${n}_$rep.json" "$out/bench_${n}_$rep.err" "$n" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
det = [json.loads(l[len("[bench detail] "):]) for l in open(sys.argv[2], errors="replace") if l.startswith("[bench detail] ")][-1]
k = det["kernel_us_per_step"]
gemm = "  ".join(f"{n.replace('disc_', '').replace('_kernel', '')} {v:6.1f}" for n, v in k.items() if n.startswith("disc_"))
r = d["roofline"]
print(f"{sys.argv[3]:28s} {d['ms_per_step']*1e3:7.1f} us/step  env {k['env_step_reference_kernel']:6.1f}  {gemm}  "
      f"tail {k.get('step_tail_kernel', 0.0):4.1f} | 8192: {r['envs_8192']['ms_per_step']*1e3:5.1f}  4096: {r['envs_4096']['ms_per_step']*1e3:5.1f}")
PY
  done
done
