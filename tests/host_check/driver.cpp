// Host-side argument validation of every C ABI entry point under ASan/UBSan: calls that return
// before any launch (empty batches, bad arguments).  No GPU needed.
#include <cstdio>
#include <cmath>
#include "sccbf.h"
int main()
{
    sccbf_params p; sccbf_default_params(&p);
    sccbf_nominal_params np; sccbf_default_nominal_params(&np);
    sccbf_rollout_params rp; sccbf_default_rollout_params(&rp);
    int bad = 0;
    auto expect = [&](int got, int want, const char* what) { if (got != want) { std::printf("%s: %d != %d\n", what, got, want); ++bad; } };
    expect(sccbf_solve_f32(&p, 0, 16, 0, 0, 0, 0, 0, 0, 0, 0), 0, "solve empty");
    expect(sccbf_solve_f64(&p, 8, 16, 0, 0, 0, 0, 0, 0, 0, 0), -1, "solve null");
    expect(sccbf_solve_f32(&p, -1, 16, 0, 0, 0, 0, 0, 0, 0, 0), -1, "solve negative A");
    expect(sccbf_solve_f32(&p, (int64_t)1 << 31, 16, 0, 0, 0, 0, 0, 0, 0, 0), -1, "solve huge A");
    expect(sccbf_solve_f32(&p, 0, 2000, 0, 0, 0, 0, 0, 0, 0, 0), -2, "solve rows");
    for (int dyn = -1; dyn <= 5; ++dyn) { p.dynamics = dyn; sccbf_solve_f32(&p, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0); sccbf_solve_f64(&p, 4, 4, 0, 0, 0, 0, 0, 0, 0, 0); }
    p.dynamics = 0;
    for (unsigned f = 0; f < 64; ++f) { p.flags = f; sccbf_solve_f32(&p, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0); sccbf_solve_sa_f32(&p, 0.1, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0, 0); }
    p.flags = 0;
    p.R[0] = -1; expect(sccbf_solve_f32(&p, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0), -3, "weight"); p.R[0] = 1;
    p.R[1] = NAN; expect(sccbf_qp_f64(&p, 0, 4, 0, 0, 0, 0, 0), -3, "weight nan"); p.R[1] = 0;
    p.lr = 0; expect(sccbf_solve_f32(&p, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0), -1, "lr"); p.lr = 1.45;
    for (int l : {0, 1, 2, 3, 4, 8, 16, 32, 64, 65}) { p.lanes_per_qp = l; sccbf_solve_f32(&p, 0, 300, 0, 0, 0, 0, 0, 0, 0, 0); }
    p.lanes_per_qp = 0;
    expect(sccbf_qp_f32(&p, 0, 4, 0, 0, 0, 0, 0), 0, "qp empty");
    expect(sccbf_partials_f32(&p, 0, 4, 0, 0, 0, 0, 0), 0, "partials empty");
    expect(sccbf_partials_f64(&p, 4, 4, 0, 0, 0, 0, 0), -1, "partials null");
    expect(sccbf_solve_sa_f64(&p, 0.0, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0, 0), -1, "sa dt");
    expect(sccbf_solve_sa_f64(&p, INFINITY, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0, 0), -1, "sa dt inf");
    expect(sccbf_nominal_f32(&np, 0, 10, 0, 0, 0, 0, 0, 0), 0, "nominal empty");
    expect(sccbf_nominal_f64(&np, 4, 0, 0, 0, 0, 0, 0, 0), -1, "nominal no course");
    expect(sccbf_nominal_f64(0, 4, 10, 0, 0, 0, 0, 0, 0), -1, "nominal null params");
    expect(sccbf_vehicle_step_f32(0, 0, 0, 1.45, 1.45, 0.5, 0.1, 0, 0), 0, "step empty");
    expect(sccbf_vehicle_step_f64(4, 0, 0, 0.0, 1.45, 0.5, 0.1, 0, 0), -1, "step lr");
    expect(sccbf_solve_step_f32(&p, 0.5, 0.1, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0), 0, "solve_step empty");
    expect(sccbf_solve_step_f64(&p, 0.5, 0.1, 8, 4, 0, 0, 0, 0, 0, 0, 0, 0), -1, "solve_step null");
    expect(sccbf_solve_step_f64(&p, -1.0, 0.1, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0), -1, "solve_step max_steer");
    expect(sccbf_solve_step_f32(0, 0.5, 0.1, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0), -1, "solve_step null params");
    rp.ticks = 3;
    expect(sccbf_rollout_f32(&p, &np, &rp, 0, 4, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0), 0, "rollout empty");
    expect(sccbf_rollout_f64(&p, &np, &rp, 4, 1025, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0), -2, "rollout slots");
    for (int m = -1; m <= 4; ++m) { rp.motion = m; sccbf_rollout_f32(&p, &np, &rp, 4, 4, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0); }
    for (int d = 0; d <= 3; ++d) { p.dynamics = d; sccbf_rollout_f64(&p, &np, &rp, 4, 4, 3000, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0); }
    p.dynamics = 0; rp.motion = 0;
    p.flags = 4u; expect(sccbf_rollout_f32(&p, &np, &rp, 4, 4, 10, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0), -1, "rollout slip io"); p.flags = 0;
    expect(sccbf_vehicle_step_kin_f32(0, 0, 0, 1, 2.9, 0.5, 0.1, 0), 0, "kin step empty");
    expect(sccbf_vehicle_step_kin_f64(4, 0, 0, 3, 2.9, 0.5, 0.1, 0), -1, "kin step mode");
    expect(sccbf_vehicle_step_kin_f64(4, 0, 0, 0, 0.0, 0.5, 0.1, 0), -1, "kin step L");
    expect(sccbf_vehicle_step_kin_f32(4, 0, 0, 2, 0.0, 0.5, 0.1, 0), -1, "kin step null");
    expect(sccbf_prepare_obs_f32(0, 4, 0, 0, 0, 0), 0, "prepare empty");
    expect(sccbf_prepare_obs_f64(4, 4, 0, 0, 0, 0), -1, "prepare null");
    expect(sccbf_prepare_obs_f32(4, 4, 0, 0, 1u | 8u, 0), -1, "prepare flags");
    expect(sccbf_ingest_bbox_f32(0, 4, 0, 0, 0.5, 0, 0, 0), 0, "ingest empty");
    expect(sccbf_ingest_bbox_f64(4, 4, 0, 0, 0.5, 0, 0, 0), -1, "ingest null");
    expect(sccbf_lane_fit_f32(0, 8, 3, 0, 0, 0, 1.5, 0, 2, 0, 0, 0), 0, "fit empty");
    expect(sccbf_lane_fit_f64(4, 8, 4, 0, 0, 0, 1.5, 0, 2, 0, 0, 0), -1, "fit degree");
    for (int c = -8; c <= 1; ++c) if (!sccbf_strerror(c)) ++bad;
    std::printf("host check: %d mismatches, version %d\n", bad, sccbf_version());
    return bad;
}
