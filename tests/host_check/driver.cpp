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
    for (unsigned f = 0; f < 256; ++f) { p.flags = f; sccbf_solve_f32(&p, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0); sccbf_solve_sa_f32(&p, 0.1, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0, 0); }
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
    // several batches per launch: device table, and records from host memory staged in the library's own table
    expect(sccbf_solve_batches_f32(&p, 0, 16, 64, 0, 0), 0, "batches none");
    expect(sccbf_solve_batches_f64(&p, 4, 16, 0, 0, 0), 0, "batches no agents");
    expect(sccbf_solve_batches_f32(&p, 4, 16, 64, 0, 0), -1, "batches no table");
    expect(sccbf_solve_batches_f64(0, 1, 16, 64, 0, 0), -1, "batches no params");
    expect(sccbf_solve_batches_f32(&p, -1, 16, 64, 0, 0), -1, "batches negative count");
    expect(sccbf_solve_batches_f32(&p, SCCBF_MAX_BATCHES + 1, 16, 64, 0, 0), -1, "batches too many");
    expect(sccbf_solve_batches_f64(&p, 4, 16, (int64_t)1 << 31, 0, 0), -1, "batches huge A_max");
    expect(sccbf_solve_batches_f32(&p, 4, 1025, 64, 0, 0), -2, "batches rows");
    for (int d = 2; d <= 3; ++d) { p.dynamics = d; expect(sccbf_solve_batches_f32(&p, 0, 16, 64, 0, 0), -1, "batches model"); }
    p.dynamics = 0;
    for (unsigned f = 0; f < 256; ++f) { p.flags = f; sccbf_solve_batches_f32(&p, 0, 4, 64, 0, 0); sccbf_solve_batches_f64(&p, 4, 4, 64, 0, 0); }
    p.flags = 0;
    {
        sccbf_batch recs[3] = {};
        static sccbf_batch many[SCCBF_MAX_STATIC_BATCHES + 1];
        expect(sccbf_solve_batches_static(&p, 2, 1, 16, recs, 0), -1, "static scalar size");
        expect(sccbf_solve_batches_static(0, 4, 1, 16, recs, 0), -1, "static no params");
        expect(sccbf_solve_batches_static(&p, 4, -1, 16, recs, 0), -1, "static negative count");
        expect(sccbf_solve_batches_static(&p, 4, SCCBF_MAX_STATIC_BATCHES + 1, 16, many, 0), SCCBF_ERR_BATCHES, "static too many");
        expect(sccbf_solve_batches_static(&p, 8, 3, 16, 0, 0), -1, "static no records");
        expect(sccbf_solve_batches_static(&p, 8, 0, 16, 0, 0), 0, "static none");
        expect(sccbf_solve_batches_static(&p, 8, 3, 16, recs, 0), 0, "static empty batches");
        expect(sccbf_solve_batches_static(&p, 8, 3, 1025, recs, 0), -2, "static rows");
        recs[1].A = 8;
        expect(sccbf_solve_batches_static(&p, 4, 3, 16, recs, 0), -1, "static agents without arrays");
        recs[1].A = -1;
        expect(sccbf_solve_batches_static(&p, 4, 3, 16, recs, 0), -1, "static negative agents");
        recs[1].A = (int64_t)1 << 31;
        expect(sccbf_solve_batches_static(&p, 4, 3, 16, recs, 0), -1, "static huge batch");
    }
    // one layout of obs per call; the least-violation pass where it is refused
    p.flags = SCCBF_FLAG_OBS_AGENT_MAJOR | SCCBF_FLAG_OBS_TILED; expect(sccbf_solve_f32(&p, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0), -1, "two layouts");
    p.flags = SCCBF_FLAG_OBS_AGENT_MAJOR | SCCBF_FLAG_OBS_SHARED; expect(sccbf_solve_f64(&p, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0), -1, "two layouts");
    p.flags = SCCBF_FLAG_OBS_AGENT_MAJOR; expect(sccbf_solve_f32(&p, 0, 4, 0, 0, 0, 0, 0, 0, 0, 0), 0, "per-agent lists");
    expect(sccbf_prepare_obs_f32(4, 4, 0, 0, SCCBF_FLAG_OBS_AGENT_MAJOR | SCCBF_FLAG_OBS_TILED, 0), -1, "prepare two layouts");
    expect(sccbf_ingest_bbox_f32(4, 4, 0, 0, 0.5, SCCBF_FLAG_OBS_AGENT_MAJOR | SCCBF_FLAG_OBS_TILED, 0, 0), -1, "ingest two layouts");
    expect(sccbf_ingest_bbox_f64(0, 4, 0, 0, 0.5, SCCBF_FLAG_OBS_AGENT_MAJOR, 0, 0), 0, "ingest lists empty");
    expect(sccbf_lane_fit_f32(4, 8, 3, 0, 0, 0, 1.5, 0, 2, SCCBF_FLAG_OBS_AGENT_MAJOR | SCCBF_FLAG_OBS_TILED, 0, 0), -1, "fit two layouts");
    p.flags = SCCBF_FLAG_INFEASIBLE_MIN_VIOLATION;
    expect(sccbf_solve_f64(&p, 4, 2, (const double*)8, (const double*)8, (const double*)8, (const uint8_t*)8, (double*)8, (int*)8, 0, 0), -1, "least violation needs rows_out");
    expect(sccbf_solve_f64(&p, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0), 0, "least violation, nothing to do");
    expect(sccbf_solve_step_f32(&p, 0.5, 0.1, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0), -1, "least violation with a step");
    expect(sccbf_solve_batches_f32(&p, 0, 2, 4, 0, 0), -1, "least violation with batches");
    p.flags = 0;
    sccbf_prewarm();   // without a device: an error code, never a crash
    for (int c = -8; c <= 1; ++c) if (!sccbf_strerror(c)) ++bad;
    std::printf("host check: %d mismatches, version %d\n", bad, sccbf_version());
    return bad;
}
