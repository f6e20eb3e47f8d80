"""Randomised parity soak of the flat search: random shapes, metrics, storage types, modes, duplicates / ties / NaN / zero rows,
adds split over several calls, id bases -- the HIP path (C-ABI) against the CPU oracle, ids and score bits.  Runs for
FUZZ_SECONDS (default 240) and prints one summary line; any mismatch is printed with its seed and ends the run with status 1.
Test infrastructure (imports oracle/): not part of the product."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from rag_amd.index import FlatIndex, MODE_AUTO, MODE_EXACT, MODE_FAST

budget = float(os.environ.get("FUZZ_SECONDS", "240"))
seed0 = int(os.environ.get("FUZZ_SEED", "1"))
t_end = time.time() + budget
cases = fails = 0
routes = {"fast": 0, "exact": 0}
it = 0
while time.time() < t_end:
    seed = seed0 * 100003 + it
    it += 1
    rng = np.random.default_rng(seed)
    d = int(rng.choice([32, 64, 100, 128, 200, 256, 384, 384, 384, 512, 768, 768]))
    big = rng.random() < 0.5
    n = int(rng.integers(33000, 260000)) if big else int(rng.integers(1, 33000))
    if d >= 512 and n > 120000:
        n = 120000
    nq = int(rng.choice([1, 2, 7, 16, 17, 33, 64, 65, 130, 257, 400]))
    k = int(rng.choice([1, 3, 5, 10, 10, 32, 100, 128]))
    metric = "ip" if rng.random() < 0.5 else "l2"
    dtype = "bf16" if rng.random() < 0.5 else "f32"
    mode = [MODE_AUTO, MODE_AUTO, MODE_FAST, MODE_EXACT][int(rng.integers(0, 4))]
    if mode == MODE_FAST and not (n >= 32768 and d <= 768 and k <= 100):
        mode = MODE_AUTO                                      # MODE_FAST is refused (documented) where the MFMA scan does not apply
    X = rng.standard_normal((n, d)).astype(np.float32)
    if rng.random() < 0.7:
        X /= np.maximum(np.linalg.norm(X, axis=1, keepdims=True), 1e-20)
    else:
        X *= np.float32(10.0 ** rng.integers(-3, 4))
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    if rng.random() < 0.7:
        Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    # planted structure: duplicates (ties), a query equal to a row, zero rows, sometimes NaN
    if n > 50:
        for _ in range(int(rng.integers(0, 4))):
            a = int(rng.integers(0, n)); m = int(rng.integers(1, min(40, n)))
            dst = rng.integers(0, n, size=m)
            X[dst] = X[a]
        Q[0] = X[int(rng.integers(0, n))]
        X[int(rng.integers(0, n))] = 0.0
        if rng.random() < 0.2:
            X[int(rng.integers(0, n)), int(rng.integers(0, d))] = np.nan
    id_base = int(rng.choice([0, 0, 12345, 3_000_000_000]))
    idx = FlatIndex(d, metric=metric, dtype=dtype)
    if id_base:
        idx.set_id_base(id_base)
    idx.set_mode(mode)
    cuts = sorted(set([0, n] + [int(c) for c in rng.integers(0, n + 1, size=int(rng.integers(0, 3)))]))
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b > a:
            idx.add(X[a:b])
    try:
        D, I = idx.search(Q, k)
    except Exception as e:                                    # a refusal is a failure here: every drawn case is in range
        print(json.dumps({"seed": seed, "error": str(e), "n": n, "d": d, "nq": nq, "k": k, "metric": metric, "dtype": dtype, "mode": mode}), flush=True)
        fails += 1
        break
    m = {"ip": oracle.METRIC_IP, "l2": oracle.METRIC_L2}[metric]
    t = {"f32": oracle.DTYPE_F32, "bf16": oracle.DTYPE_BF16}[dtype]
    Do, Io = oracle.search(X, Q, k, metric=m, dtype=t, id_base=id_base)
    st = idx.stats()
    routes["fast"] += st["fast"]; routes["exact"] += st["exact"]
    cases += 1
    if cases % 20 == 0:
        print(json.dumps({"progress": cases, "queries_by_route": routes}), flush=True)
    if not (np.array_equal(I, Io) and np.array_equal(D.view(np.uint32), Do.view(np.uint32))):
        bad = np.argwhere((I != Io) | (D.view(np.uint32) != Do.view(np.uint32)))
        print(json.dumps({"seed": seed, "mismatch_at": bad[:5].tolist(), "n": n, "d": d, "nq": nq, "k": k, "metric": metric,
                          "dtype": dtype, "mode": mode, "id_base": id_base}), flush=True)
        fails += 1
        break
    del idx
print(json.dumps({"cases": cases, "failures": fails, "queries_by_route": routes, "seconds": round(budget - (t_end - time.time()), 1), "seed0": seed0}), flush=True)
sys.exit(1 if fails else 0)
