# bench.py over the tuning switches of the library; one JSON line per combination under gpurun_out/
set -e
out=gpurun_out/exp_matrix.txt
: > $out
for lanes in 4 8; do for incr in 0 1; do for nt in 0 1; do
  SCCBF_INCR=$incr SCCBF_NT=$nt timeout -k 10 200 python bench.py --no-cpu --no-large --no-fp64 --lanes $lanes --steps 10000 --warmup 2000 > gpurun_out/em.json 2>> gpurun_out/em.err
  python - "$lanes" "$incr" "$nt" >> $out <<'PY'
import json, sys
d = json.load(open("gpurun_out/em.json"))
print("lanes %s incr %s nt %s  us/step %.3f  frac %.4f  infeasible %.4f" % (sys.argv[1], sys.argv[2], sys.argv[3], d["roofline"]["kernel_ms"] * 1e3, d["roofline"]["frac"], d["infeasible_fraction"]))
PY
done; done; done
