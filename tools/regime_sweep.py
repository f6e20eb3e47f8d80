"""Regime sweep of the flat-search hot path on one MI355X (not a bench line; evidence for DESIGN.md):
query-batch sizes from the app's Q=1 to BASELINE's 10k, the fp32 config, the 768-d / k=100 config.
Prints one JSON object per case.  Usage: python tools/regime_sweep.py [--rows N] [--cases a,b,c,d]"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag_amd.index import FlatIndex  # noqa: E402


def make_queries(d, Q, dev):
    g = FlatIndex(d, metric="ip", dtype="f32")
    g.add_synthetic(Q, seed=1, normalize=True)
    q = torch.from_numpy(g.reconstruct_n(0, Q)).to(dev)
    del g
    return q


def run_case(name, d, N, dtype, metric, Qs, k, reps, dev):
    idx = FlatIndex(d, metric=metric, dtype=dtype)
    idx.reserve(N)
    idx.add_synthetic(N, seed=0, normalize=True)
    qall = make_queries(d, max(Qs), dev)
    esz = 2 if dtype == "bf16" else 4
    for Q in Qs:
        q = qall[:Q].contiguous()
        idx.search_torch(q, k)
        torch.cuda.synchronize()
        wall, scan = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            idx.search_torch(q, k)
            torch.cuda.synchronize()
            wall.append(time.perf_counter() - t0)
            scan.append(idx.last_scan_time())
        w = float(np.median(wall))
        s = float(np.median([x["ms"] for x in scan])) * 1e-3
        st = idx.stats()
        print(json.dumps({"case": name, "rows": N, "dim": d, "dtype": dtype, "metric": metric, "Q": Q, "k": k,
                          "wall_ms": round(w * 1e3, 3), "scan_ms": round(s * 1e3, 3), "scan_launches": scan[-1]["launches"],
                          "kind": scan[-1]["kind"], "qps": round(Q / w, 1),
                          "scan_GBs": round(N * d * esz / s / 1e9, 1), "scan_TFLOPs": round(2.0 * Q * N * d / s / 1e12, 2),
                          "wall_GBs": round(N * d * esz / w / 1e9, 1), "exact_reruns_total": st["exact"]}), flush=True)
    del idx
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=20_000_000)
    ap.add_argument("--cases", default="a,b,c,d")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    cases = set(args.cases.split(","))
    if "a" in cases:
        run_case("bf16_ip_384_qsweep", 384, args.rows, "bf16", "ip", [1, 8, 32, 128, 256, 1024, 10000], 10, args.reps, dev)
    if "b" in cases:
        run_case("f32_ip_384_config2", 384, 1_000_000, "f32", "ip", [1, 1000], 10, args.reps, dev)
        run_case("f32_l2_384_default_store", 384, 1_000_000, "f32", "l2", [1, 1000], 5, args.reps, dev)
    if "c" in cases:
        run_case("bf16_ip_768_k100_config5", 768, args.rows // 4, "bf16", "ip", [1, 256, 10000], 100, args.reps, dev)
    if "d" in cases:
        run_case("bf16_l2_384", 384, 1_000_000, "bf16", "l2", [1, 1000], 10, args.reps, dev)


if __name__ == "__main__":
    main()
