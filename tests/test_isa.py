"""What the compiler made of the bench's kernel, checked on the gfx950 assembly (no GPU needed).

The kernel's overlap of loads and arithmetic rests on the compiler's counted `s_waitcnt vmcnt(N)`
waits (loads issued in consumption order, every row waiting only for its own slots); a load in a
conditional path, or another order of the loads, silently turns them into vmcnt(0).  Register and
scratch use decide how many wavefronts fit on a SIMD."""
import os
import re
import shutil
import subprocess

import pytest

from sccav_cbf_amd import _build

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


KERNELS = """
#include "sccbf_loop.hip.h"
#include "sccbf_batches.hip.h"
namespace {
#define GROUP(T, G, RPL, NT) template __global__ void k_group<T, G, RPL, false, NT>( \\
    long long, const T*, const T*, const T*, const unsigned char*, int, unsigned, const KArgs<T>);
GROUP(float, 4, 4, false)
GROUP(float, 4, 4, true)
GROUP(float, 1, 16, false)
GROUP(float, 16, 1, false)
GROUP(double, 4, 4, false)
GROUP(double, 4, 4, true)
GROUP(float, 8, 16, true)
template __global__ void k_group_batches<float, 4, 4, true>(const sccbf_batch*, int, unsigned, const KArgs<float>);
template __global__ void k_group_batches<double, 4, 4, true>(const sccbf_batch*, int, unsigned, const KArgs<double>);
template __global__ void k_nominal<float>(const NArgs<float>);
}
"""


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    """The kernels under test, instantiated from the library's headers with the library's flags (the whole
    library takes minutes to compile; these take seconds and come out the same)."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    d = tmp_path_factory.mktemp("isa")
    src, out = d / "kernels.hip", d / "kernels.s"
    src.write_text(KERNELS)
    flags = [f for f in _build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    cmd = [HIPCC] + flags + ["--cuda-device-only", "-S", "-I" + os.path.dirname(_build.HDR), "-I" + _build.CSRC,
                             "-o", str(out), str(src)]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout[-2000:]
    return out.read_text()


def _kernel(asm, mangled_prefix):
    m = re.search(r"^(%s\w*):[^\n]*\n(.*?)s_endpgm" % mangled_prefix, asm, flags=re.S | re.M)
    assert m, "kernel %s not found" % mangled_prefix
    meta = re.search(r"\.amdhsa_kernel %s\n(.*?)\.end_amdhsa_kernel" % re.escape(m.group(1)), asm, flags=re.S)
    assert meta
    return m.group(2), meta.group(1)


def _meta(meta, key):
    return int(re.search(r"\.amdhsa_%s (\d+)" % key, meta).group(1))


@pytest.mark.parametrize("nt", ["Lb0", "Lb1"], ids=["cached", "non-temporal"])
def test_bench_kernel_keeps_counted_waits_and_fits_seven_waves(asm, nt):
    # k_group<float, 4, 4, false, NT>: 4 lanes per QP, 4 rows per lane: 2 + 4 * 3 = 14 loads per lane
    # (the bench sets the streamed flag: the non-temporal instantiation is the one it runs)
    body, meta = _kernel(asm, "_ZN12_GLOBAL__N_17k_groupIfLi4ELi4ELb0E%sEEEv" % nt)
    waits = [int(n) for n in re.findall(r"s_waitcnt vmcnt\((\d+)\)", body)]
    assert max(waits) >= 12, waits                    # the ego arithmetic starts with 12-13 loads still in flight
    assert len(set(waits)) >= 12, sorted(set(waits))  # and every later row waits for its own loads only
    assert _meta(meta, "private_segment_fixed_size") == 0          # no scratch
    assert _meta(meta, "next_free_vgpr") <= 72                     # 7 wavefronts per SIMD
    assert _meta(meta, "user_sgpr_kernarg_preload_length") >= 12   # leading arguments preloaded
    assert 12000 <= _meta(meta, "group_segment_fixed_size") <= 14000   # the row table: 3 x 64 x 17 floats


def test_other_variants_have_no_scratch(asm):
    for name in ("_ZN12_GLOBAL__N_17k_groupIfLi1ELi16ELb0ELb0EEEv", "_ZN12_GLOBAL__N_17k_groupIfLi16ELi1ELb0ELb0EEEv",
                 "_ZN12_GLOBAL__N_17k_groupIdLi4ELi4ELb0ELb0EEEv", "_ZN12_GLOBAL__N_19k_nominalIfEEv"):
        _, meta = _kernel(asm, name)
        assert _meta(meta, "private_segment_fixed_size") == 0, name


def test_fp64_rollout_fits_two_waves_without_scratch(tmp_path):
    # the fp64 rollout as its part of the library is compiled (no machine-level hoisting of loop invariants):
    # k_rollout<double, 4, false, 4> within 256 registers per lane, at most four spilled (it was 384 B with the hoisting)
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    src = tmp_path / "rollout_f64.hip"
    src.write_text('#include "sccbf_loop.hip.h"\nnamespace {\n'
                   'template __global__ void k_rollout<double, 4, false, 4>(const RArgs<double>);\n}\n')
    extra = dict(_build.PARTS)["ROLLOUT_F64"]
    flags = [f for f in _build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")] + extra
    out = tmp_path / "rollout_f64.s"
    cmd = [HIPCC] + flags + ["--cuda-device-only", "-S", "-I" + os.path.dirname(_build.HDR), "-I" + _build.CSRC,
                             "-o", str(out), str(src)]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout[-2000:]
    _, meta = _kernel(out.read_text(), "_ZN12_GLOBAL__N_19k_rolloutIdLi4ELb0ELi4EEE")
    assert _meta(meta, "private_segment_fixed_size") <= 32
    assert _meta(meta, "next_free_vgpr") <= 256


def test_batches_kernels_keep_the_occupancy_of_the_single_launch_kernels(asm):
    # the kernel of the bench's timed region (twenty batches per launch): as k_group<float, 4, 4>, seven wavefronts per
    # SIMD, no scratch, counted waits (a loop over several turns per workgroup cost 35 registers and was dropped)
    body, meta = _kernel(asm, "_ZN12_GLOBAL__N_115k_group_batchesIfLi4ELi4ELb1EEE")
    waits = [int(n) for n in re.findall(r"s_waitcnt vmcnt\((\d+)\)", body)]
    assert max(waits) >= 12 and len(set(waits)) >= 12, sorted(set(waits))
    assert _meta(meta, "private_segment_fixed_size") == 0 and _meta(meta, "next_free_vgpr") <= 72


def test_fp64_kernels_fit_four_waves_without_scratch(asm):
    # fp64 with four rows per lane: 128 registers = four wavefronts per SIMD (a 65536-agent launch is exactly four per
    # SIMD); the device library's sincos / tan / sqrt / division used to put it at 130
    for name in ("_ZN12_GLOBAL__N_17k_groupIdLi4ELi4ELb0ELb0EEEv", "_ZN12_GLOBAL__N_17k_groupIdLi4ELi4ELb0ELb1EEEv",
                 "_ZN12_GLOBAL__N_115k_group_batchesIdLi4ELi4ELb1EEE"):
        _, meta = _kernel(asm, name)
        assert _meta(meta, "private_segment_fixed_size") == 0, name
        assert _meta(meta, "next_free_vgpr") <= 128, name


def test_sixteen_rows_per_lane_stay_inside_the_instruction_cache(asm):
    # k_group<float, 8, 16>, the kernel of a 128-row launch, unrolls sixteen rows: everything a row inlines is there sixteen
    # times.  With the degree-6 closest-point loop inlined it grew past the 64 KB instruction cache and a launch went
    # from 59 to 306 us; it is a call there (inlined only where at most four rows are unrolled).
    body, meta = _kernel(asm, "_ZN12_GLOBAL__N_17k_groupIfLi8ELi16ELb0ELb1EEEv")
    insts = [ln for ln in body.splitlines() if ln.startswith("\t") and ln.strip() and not ln.strip().startswith((".", ";"))]
    assert len(insts) <= 9800, len(insts)
    assert "lane6_closest_x_call_" in body
    assert _meta(meta, "private_segment_fixed_size") == 0
