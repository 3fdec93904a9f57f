# Build the experiment harness with -save-temps under /tmp/isa and print, per k_group variant
# given as mangled template args (e.g. Li4ELi4ELb0ELi1), the vmcnt waits, VGPRs and VALU line count.
set -e
mkdir -p /tmp/isa && cd /tmp/isa && rm -f *.bc *.o *.hipi *.out* *.s
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops $EXTRA -I/root/repo/include -save-temps -o /root/repo/tools/kexp /root/repo/tools/kexp.hip 2>&1 | grep -E "error" || true
S=kexp-hip-amdgcn-amd-amdhsa-gfx950.s
for k in "$@"; do
  n=$(grep -n "^_ZN12_GLOBAL__N_17k_groupIf${k}EEEv[A-Za-z0-9_]*:" $S | head -1 | cut -d: -f1)
  e=$(awk -v n=$n 'NR>n && /s_endpgm/ {print NR; exit}' $S)
  awk -v a=$n -v b=$e 'NR>=a && NR<=b' $S > k_$k.s
  echo "== $k: $(grep -c '^\sv_' k_$k.s) VALU lines, $(grep -c '^\ss_' k_$k.s) SALU lines, $(grep -A30 "amdhsa_kernel _ZN12_GLOBAL__N_17k_groupIf${k}E" $S | grep -E 'next_free_vgpr' | tr -d '\t')"
  grep -o "vmcnt([0-9]*)" k_$k.s | tr '\n' ' '; echo
done
