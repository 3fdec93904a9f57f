/*
 * sanitize_check.c - drives the oracle's entry points on random data under
 * AddressSanitizer and UndefinedBehaviorSanitizer (host only; `make sanitize`).
 * TEST INFRASTRUCTURE ONLY, like the oracle itself.
 */
#include "cbf_oracle.c"

#include <stdio.h>

static double urand(unsigned* s)
{
    *s = *s * 1664525u + 1013904223u;
    return (double)(*s >> 8) / 16777216.0;
}

int main(void)
{
    enum { A = 257, M = 19, N = 300 };
    unsigned seed = 12345u;
    orc_params p;
    memset(&p, 0, sizeof(p));
    p.alpha = 1.0; p.lr = 1.45; p.lf = 1.45; p.L = 2.9;
    p.R[0] = 0.5; p.R[3] = 0.5; p.zero_tol = 1e-3;
    double* state = malloc(sizeof(double) * 4 * A);
    double* u_ref = malloc(sizeof(double) * 2 * A);
    double* obs = malloc(sizeof(double) * 8 * M * A);
    uint8_t* ty = malloc(M);
    double* u = malloc(sizeof(double) * 2 * A);
    int32_t* st = malloc(sizeof(int32_t) * A);
    double* rows = malloc(sizeof(double) * 4 * M * A);
    double* slip = calloc(2 * A, sizeof(double));
    double* traj = malloc(sizeof(double) * 4 * N);
    int32_t* tgt = calloc(A, sizeof(int32_t));
    double* pid = calloc(2 * A, sizeof(double));
    double* beta = malloc(sizeof(double) * A);
    for (int a = 0; a < A; ++a) {
        state[4 * a] = 100 * urand(&seed) - 50; state[4 * a + 1] = 30 * urand(&seed);
        state[4 * a + 2] = 6.2 * urand(&seed) - 3.1; state[4 * a + 3] = 15 * urand(&seed);
        u_ref[2 * a] = 4 * urand(&seed) - 2; u_ref[2 * a + 1] = urand(&seed) - 0.5;
    }
    state[2] = 1e17; state[6] = INFINITY; state[4 * 9 + 3] = NAN;   /* diverged and non-finite agents */
    for (int m = 0; m < M; ++m) {
        static const uint8_t kinds[5] = {0, 1, 2, 3, 255};
        ty[m] = kinds[m % 5];
        for (int h = 0; h < 2; ++h)
            for (int a = 0; a < A; ++a)
                for (int c = 0; c < 4; ++c) {
                    double v = 40 * urand(&seed) - 20;
                    if (ty[m] == 2) v = (h == 0) ? (c == 0 ? 20.0 : 0.01 * (urand(&seed) - 0.5)) : (c == 0 ? 1.5 : 0.0);
                    if ((ty[m] == 0 || ty[m] == 3) && h == 0 && c >= 2) v = 2.0 + 5 * urand(&seed);
                    obs[(((size_t)m * 2 + h) * A + a) * 4 + c] = v;
                }
    }
    for (int i = 0; i < N; ++i) {
        traj[4 * i] = 0.3 * i; traj[4 * i + 1] = 15 + 5 * sin(0.02 * i);
        traj[4 * i + 2] = atan(0.1 * cos(0.02 * i) / 0.3); traj[4 * i + 3] = 8.0;
    }
    int rc = 0;
    for (int dyn = 0; dyn <= 2; ++dyn) {
        p.dynamics = dyn;
        for (int inc = 0; inc <= 1; ++inc) rc |= orc_solve_batch(&p, A, M, state, u_ref, obs, ty, u, st, rows, inc);
    }
    p.dynamics = ORC_DYN_SADBM;
    p.flags = ORC_FLAG_CONE_BETA_STATE;
    rc |= orc_solve_sa_batch(&p, 0.1, A, M, state, slip, u_ref, obs, ty, u, st, rows, 0);
    p.flags = 0; p.dynamics = 0;
    rc |= orc_qp_batch(&p, A, M, rows, u_ref, u, st, 0);
    rc |= orc_qp_batch(&p, A, M, rows, u_ref, u, st, 1);
    rc |= orc_solve_batch(&p, A, 0, state, u_ref, obs, ty, u, st, NULL, 0);     /* no rows */
    orc_nominal_batch(A, N, state, traj, tgt, pid, 2.9, 0.5, 0.01, 1.0, 0.1, 0.02, 0.1, 8.0, u);
    orc_nominal_batch(A, 1, state, traj, tgt, NULL, 2.9, 0.5, 0.0, 1.0, 0.0, 0.0, 0.1, 8.0, u);
    orc_vehicle_step_batch(A, state, u, 1.45, 1.45, 0.5236, 0.1, beta);
    printf("sanitize_check: rc %d\n", rc);
    free(state); free(u_ref); free(obs); free(ty); free(u); free(st); free(rows); free(slip);
    free(traj); free(tgt); free(pid); free(beta);
    return rc;
}
