#!/usr/bin/env python3
"""Randomised campaign for the device-side lane fit and bounding-box ingestion (fp64 and fp32)
against the oracle: random point counts, degrees, weights with fixed points, skipped points,
exact duplicates, too few points; random boxes, kinds and layouts.

    python tests/campaigns/fuzz_ingest.py [--cases 300] [--seed 0]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as orc  # noqa: E402
from sccav_cbf_amd.batched import planes_to_tiled  # noqa: E402
from sccav_cbf_amd.ingest import ingest_bbox, lane_fit  # noqa: E402

DEV = torch.device("cuda", 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    orc.build()
    misses, fits, nans = 0, 0, 0
    for case in range(args.cases):
        A = int(rng.integers(1, 300))
        K = int(rng.integers(1, 48)) if rng.random() < 0.8 else int(rng.integers(48, 140))
        deg = int(rng.integers(0, 4))
        M = int(rng.integers(1, 6))
        slot = int(rng.integers(0, M))
        tiled = bool(rng.integers(0, 2))
        x0 = rng.uniform(-200, 200, A)
        px = x0[None, :] + np.sort(rng.uniform(-10, 80, (K, A)), axis=0)
        px = np.round(px, 2)                                    # exact duplicates happen
        coef = rng.normal(0, [5, 0.3, 0.01, 2e-4], (A, 4))
        py = sum(coef[None, :, j] * (px - x0[None, :]) ** j for j in range(4)) + rng.normal(0, 0.1, (K, A))
        sigma = np.full((K, A), 10.0)
        sigma[rng.uniform(size=(K, A)) < 0.1] = 0.01           # fixed points
        sigma[rng.uniform(size=(K, A)) < 0.15] = 0.0            # skipped points
        for dtype, tol in ((torch.float64, 1e-8), (torch.float32, 5e-2)):
            x, y, s = px, py, sigma
            if dtype == torch.float32:
                x, y, s = (v.astype(np.float32).astype(np.float64) for v in (px, py, sigma))
            used = s > 0
            distinct = np.array([np.unique(x[used[:, a], a]).size for a in range(A)])
            expect_nan = distinct < deg + 1
            # an exactly determined fit (as many distinct abscissae as coefficients, some a few centimetres
            # apart) is an interpolation through noise: judged in fp64 only, to 1e-6
            okc = ~expect_nan & (distinct >= deg + 2)
            okl = ~expect_nan & (distinct == deg + 1) & (dtype == torch.float64)
            exp = np.zeros((A, 4))
            if (~expect_nan).any():
                sel = np.where(~expect_nan)[0]
                exp[sel] = orc.lane_fit(x[:, sel], y[:, sel], s[:, sel], degree=deg)
            t = lambda v: torch.from_numpy(np.ascontiguousarray(v)).to(DEV).to(dtype)      # noqa: E731
            obs = torch.zeros((M, 2, A, 4), dtype=dtype, device=DEV)
            if tiled:
                obs = planes_to_tiled(obs)
            lane_fit(t(x), t(y), obs, slot, sigma=t(s), degree=deg, buffer=1.5)
            torch.cuda.synchronize()
            if tiled:       # tiles (ceil(A/64), M, 2, 64, 4) -> planes
                o = obs.double().cpu().numpy()
                o = o.transpose(1, 2, 0, 3, 4).reshape(M, 2, -1, 4)[:, :, :A, :]
            else:
                o = obs.double().cpu().numpy()
            c = o[slot, 0]
            got_nan = np.isnan(c).any(axis=1)
            fits += int((~expect_nan).sum())
            nans += int(expect_nan.sum())
            if (got_nan != expect_nan).any():
                misses += 1
                a = int(np.argmax(got_nan != expect_nan))
                print("MISS nan case %d %s A=%d K=%d deg=%d: agent %d got_nan %d expected %d distinct %d"
                      % (case, dtype, A, K, deg, a, got_nan[a], expect_nan[a], distinct[a]), flush=True)
                continue
            if okc.any() or okl.any():
                lo = np.where(used, x, np.inf).min(axis=0)
                hi = np.where(used, x, -np.inf).max(axis=0)
                xs = np.linspace(lo, hi, 16)
                val = lambda cc: sum(cc[None, :, j] * xs ** j for j in range(4))            # noqa: E731
                d = (np.abs(val(c) - val(exp)) / (1 + np.abs(val(exp))))[:, okc].max() if okc.any() else 0.0
                if d > tol:
                    misses += 1
                    print("MISS fit case %d %s A=%d K=%d deg=%d: %.3g" % (case, dtype, A, K, deg, d), flush=True)
                if okl.any():
                    d = (np.abs(val(c) - val(exp)) / (1 + np.abs(val(exp))))[:, okl].max()
                    if d > 1e-6:
                        misses += 1
                        print("MISS exact fit case %d A=%d K=%d deg=%d: %.3g" % (case, A, K, deg, d), flush=True)
            if not (np.allclose(o[slot, 1], np.array([1.5, 0, 0, 0])[None, :]) and
                    all((o[m] == 0).all() for m in range(M) if m != slot)):
                misses += 1
                print("MISS slot case %d: other slots touched or buffer wrong" % case, flush=True)
        # boxes
        Mb = int(rng.integers(1, 20))
        bbox = rng.uniform(-50, 50, (Mb, 2, A, 4))
        bbox[:, 0, :, 2:4] = rng.uniform(0.5, 4, (Mb, A, 2))
        per_agent = bool(rng.integers(0, 2))
        kind = rng.choice(np.array([0, 1, 255], dtype=np.uint8), (Mb, A) if per_agent else (Mb,))
        buf = float(rng.uniform(0, 2))
        exp = orc.ingest_bbox(bbox, kind, buffer=buf)
        for dtype, tol in ((torch.float64, 1e-13), (torch.float32, 1e-5)):
            b = torch.from_numpy(bbox).to(DEV).to(dtype)
            k = torch.from_numpy(kind).to(DEV)
            got = ingest_bbox(b, k, buffer=buf)
            til = ingest_bbox(b, k, buffer=buf, tiled=True)
            torch.cuda.synchronize()
            if np.abs(got.double().cpu().numpy() - exp).max() > tol * 100 or not torch.equal(planes_to_tiled(got), til):
                misses += 1
                print("MISS bbox case %d %s" % (case, dtype), flush=True)
        if case % 100 == 99:
            print("... %d cases done, %d misses" % (case + 1, misses), flush=True)
    print("cases %d: fits %d, underdetermined %d, misses %d" % (args.cases, fits, nans, misses))
    sys.exit(1 if misses else 0)


if __name__ == "__main__":
    main()
