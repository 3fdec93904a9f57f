# gpurun_out/r3_* (written by tools/round_profile.sh on the GPU box) -> profiles/r03_* and traffic*.json
set -e
cd "$(dirname "$0")/.."
G=gpurun_out
cp $G/r3_traffic.json profiles/traffic.json
cp $G/r3_traffic_large.json profiles/traffic_large.json
cp $G/r3_stats_kernel_stats.csv profiles/r03_kernel_stats.csv
cp $G/r3_stats_single_kernel_stats.csv profiles/r03_kernel_stats_one_launch_per_step.csv
cp $G/r3_large_stats_kernel_stats.csv profiles/r03_large_kernel_stats.csv
cp $G/r3_f64_kernel_stats.csv profiles/r03_f64_kernel_stats.csv
cp $G/r3_f64_pmc.txt profiles/r03_f64_pmc.txt
cp $G/r3_rows_sweep.txt profiles/r03_rows_sweep.txt
cp $G/r3_rows_sweep_f64.txt profiles/r03_rows_sweep_f64.txt
[ -s $G/r3_rows_fused.txt ] && cp $G/r3_rows_fused.txt profiles/r03_rows_fused.txt
(for f in fsize write sq; do echo "== $f"; cat $G/r3_pmc_$f.txt; done; for f in large_fsize large_write; do echo "== $f"; cat $G/r3_$f.txt; done) > profiles/r03_pmc_summary.txt
(for L in agents tiles; do for M in 128 257 1024; do echo "== layout $L, $M rows"; cat $G/r3_rows_pmc_${L}_$M.txt; done; done) > profiles/r03_rows_pmc.txt
[ -s $G/r3_bench.json ] && cp $G/r3_bench.json profiles/r03_bench.json
[ -s $G/r3_bench_steps20.json ] && cp $G/r3_bench_steps20.json profiles/r03_bench_steps20.json
