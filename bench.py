#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json `metric`):

    queries/sec, exact top-10 over a 100M x 384 bf16 corpus (IP on unit vectors = cosine),
    10k-query batch, corpus row-sharded over N GPUs, per-GPU top-k + all-gather + merge.

A "step" is one pass of the whole query batch over the whole (sharded) corpus, inputs
resident in HBM.  `python bench.py --gpus N --steps K --warmup W`; for N > 1 launch under
torch.distributed.run (one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

Strong scaling: the corpus size is fixed (the metric is quoted at 100M rows), each of the N
ranks holds rows [r*N/G, (r+1)*N/G).  The only exchange step is ONE all-gather of the
per-rank packed (score, id) blocks, Q*k*12 bytes per rank (rag_amd.distributed).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def _sha256(paths):
    """SHA-256 of one file, or of several concatenated in the order given."""
    import hashlib
    h = hashlib.sha256()
    for path in ([paths] if isinstance(paths, str) else paths):
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def pmc_traffic(name, **case):
    """HBM-side bytes per launch from the committed rocprofv3 PMC pass (profiles/scan_traffic.json; counters cannot be
    read from inside the timed process).  The file records the SHA-256 of the kernel source each pass was taken at:
    when the source in this tree differs, the number describes another build and None is returned."""
    try:
        with open(os.path.join(ROOT, "profiles", "scan_traffic.json")) as f:
            doc = json.load(f)
        sec = doc[name]
        srcs = sec["kernel_source"]
        if _sha256([os.path.join(ROOT, p) for p in ([srcs] if isinstance(srcs, str) else srcs)]) != sec["kernel_source_sha256"]:
            return None
        for c in sec["cases"]:
            if all(c.get(k) == v for k, v in case.items()):
                return c["traffic_GB_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def bench_embed(args, dev, rank, world, barrier):
    """Second half of BASELINE.json's metric: chunks/s, bulk embedding of 256-token chunks with a MiniLM-L6 encoder
    (seeded random weights, synthetic token ids), data-parallel, no collective.  Three measurements:
      value     ids and mask resident in HBM, all-ones mask (the peak)
      via_api   the same chunks from HOST token ids to HOST embeddings through mi_encoder_encode_ragged, the call
                MI355Embeddings.embed_tokenized / embed_documents_array makes (FAISS.from_documents, advanced_rag.py:295)
      ragged    lengths uniform in [32, 256] through the same call (the realistic mix)
    plus `rerank`: cross-encoder pairs/s through mi_encoder_classify_ragged (BASELINE config 5's consumer)."""
    from rag_amd.embeddings import Encoder
    from rag_amd.encoder_weights import minilm_l6_config, random_blob
    cfg = minilm_l6_config()
    blob = random_blob(cfg, seed=0, init_range=0.02)
    enc = Encoder(cfg, blob, device=dev.index)
    B, L = args.embed_chunks, 256
    H, I, nl = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    ids_h = torch.randint(1000, 30000, (B, L), generator=g, dtype=torch.int32)
    ids = ids_h.to(dev)
    mask = torch.ones((B, L), dtype=torch.int32, device=dev)
    enc.encode_ids_torch(ids[:256], mask[:256])          # warm-up

    def timed(fn, steps):
        barrier()
        t0 = time.perf_counter()
        out = None
        for _ in range(steps):
            out = fn()
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, out

    def flops(lens):                                     # algorithmic: per token 8 h^2 + 4 h f + 4 L h, per layer
        lens = np.asarray(lens, dtype=np.float64)
        return float(nl * np.sum(lens * (8.0 * H * H + 4.0 * H * I + 4.0 * lens * H)))

    def roof(fl, el):
        ach = fl / el / 1e12
        return {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4)}

    el, E = timed(lambda: enc.encode_ids_torch(ids, mask), args.embed_steps)
    chunks = B * args.embed_steps * world
    rl = roof(flops([L] * B) * args.embed_steps, el)
    rl.update({"traffic": pmc_traffic("encoder", chunks=B, tokens=L),
               "note": "whole encoder forward (all kernels), algorithmic 6.04 GFLOP per chunk; traffic = FETCH + WRITE bytes "
                       "per 4096-chunk call from the committed PMC pass (GB), null when encoder.hip / encoder_ffn3.hip changed since"})
    res = {"metric": "chunks/s embed (256-token chunks, MiniLM-L6, bf16 MFMA)", "value": round(chunks / el, 1),
           "unit": "chunks/s", "chunks_per_step_per_gpu": B, "steps": args.embed_steps, "roofline": rl, "cpu_baseline": None}

    # ---- through the boundary the application calls: host CSR token ids in, host embeddings out
    flat = ids_h.numpy().reshape(-1)
    offsets = np.arange(B + 1, dtype=np.int64) * L
    enc.encode_ragged(flat[:64 * L], offsets[:65], max_len=256)
    st0 = enc.stats()
    el_api, E_api = timed(lambda: enc.encode_ragged(flat, offsets, max_len=256), args.embed_steps)
    st1 = enc.stats()
    res["via_api"] = {"value": round(chunks / el_api, 1), "unit": "chunks/s", "frac_of_resident": round(el / el_api, 4),
                      "bulk_passes": st1["bulk_passes"] - st0["bulk_passes"], "latency_passes": st1["latency_passes"] - st0["latency_passes"],
                      "path": "host int32 token ids (CSR) -> mi_encoder_encode_ragged -> host fp32 embeddings; PCIe both ways, "
                              "length sort, token-budget passes and pinned staging inside the timed region",
                      "max_abs_diff_vs_resident": float(np.abs(E_api - E.cpu().numpy()).max())}

    # ---- ragged mix: lengths uniform in [32, 256]
    rng = np.random.default_rng(77 + rank)
    lens = rng.integers(32, 257, size=B)
    roff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    rflat = rng.integers(1000, 30000, size=int(roff[-1])).astype(np.int32)
    enc.encode_ragged(rflat[:roff[64]], roff[:65], max_len=256)
    st0 = enc.stats()
    el_r, _ = timed(lambda: enc.encode_ragged(rflat, roff, max_len=256), args.embed_steps)
    st1 = enc.stats()
    real_tok = int(lens.sum()) * args.embed_steps
    res["ragged"] = {"value": round(chunks / el_r, 1), "unit": "chunks/s", "lengths": "uniform in [32, 256], mean %.1f" % lens.mean(),
                     "tokens_per_s": round(real_tok * world / el_r, 1),
                     "padding_overhead": round((st1["tokens"] - st0["tokens"]) / real_tok - 1.0, 4),
                     "bulk_passes": st1["bulk_passes"] - st0["bulk_passes"], "latency_passes": st1["latency_passes"] - st0["latency_passes"],
                     "roofline": roof(flops(lens) * args.embed_steps, el_r)}

    # ---- BASELINE config 5's consumer: top-100 lists -> cross-encoder scores (ms-marco-MiniLM-L-6 dimensions, seeded random
    # weights): 32 queries x 100 (query, passage) pairs of 64..384 tokens through mi_encoder_classify_ragged (host in, host out)
    if B >= 256:
        from rag_amd.encoder_weights import random_head
        enc.set_head(*random_head(cfg, num_labels=1, seed=5, init_range=0.1))
        prng = np.random.default_rng(99 + rank)
        plen = prng.integers(64, 385, size=3200)
        poff = np.concatenate([[0], np.cumsum(plen)]).astype(np.int64)
        pflat = prng.integers(1000, 30000, size=int(poff[-1])).astype(np.int32)
        ptype = np.concatenate([(np.arange(m) >= 16).astype(np.int32) for m in plen])
        enc.classify_ragged(pflat[:poff[100]], ptype[:poff[100]], poff[:101])
        el_p, _ = timed(lambda: enc.classify_ragged(pflat, ptype, poff), args.embed_steps)
        res["rerank"] = {"value": round(3200 * args.embed_steps * world / el_p, 1), "unit": "pairs/s",
                         "pairs": "32 queries x 100 candidates, 64..384 tokens per pair (mean %.0f), packed passes" % plen.mean(),
                         "tokens_per_s": round(float(plen.sum()) * args.embed_steps * world / el_p, 1),
                         "roofline": roof(flops(plen) * args.embed_steps, el_p)}

    # ---- from STRINGS, what advanced_rag.py:295 passes: a local model directory (seeded weights + a synthetic WordPiece
    # vocabulary, written here: nothing can be downloaded), B texts of ~1,700 characters that tokenise to 256 word pieces
    # after truncation, through HuggingFaceEmbeddings.embed_documents_array (tokenise slab i + 1 on a worker thread while the
    # GPU has slab i).  Rank 0 only: the tokenizer uses the host's cores, which the ranks of one box share.
    if rank == 0 and B >= 256:
        import shutil
        import tempfile
        from rag_amd.embeddings import MI355Embeddings
        from rag_amd.encoder_weights import synthetic_texts, write_synthetic_model_dir
        mdir = tempfile.mkdtemp(prefix="mi355rag_bench_model_")
        try:
            write_synthetic_model_dir(mdir, cfg, blob)
            emb = MI355Embeddings(model_name=mdir, model_kwargs={"device": dev.index}, max_seq_length=L, slab_texts=1024)
            texts = synthetic_texts(B, 215, cfg.vocab_size, seed=3)
            emb.embed_documents_array(texts[:256])
            t0 = time.perf_counter()
            Et = emb.embed_documents_array(texts)
            el_t = time.perf_counter() - t0
            tm = dict(emb.last_timing)
            t0 = time.perf_counter()
            emb._tokenize_csr(texts[:1024])
            tok_rate = 1024 / (time.perf_counter() - t0)
            # the reference library on the same texts, and that the native tokenizer returns its tokens
            t0 = time.perf_counter()
            ref_ids = [e.ids for e in (getattr(emb.tokenizer, "encode_batch_fast", None) or emb.tokenizer.encode_batch)(texts[:1024])]
            ref_rate = 1024 / (time.perf_counter() - t0)
            nat = emb.native_tokenizer is not None
            tok_equal = None
            if nat:
                f_, o_ = emb._tokenize_csr(texts[:1024])
                tok_equal = all(f_[o_[i]:o_[i + 1]].tolist() == ref_ids[i] for i in range(1024))
            res["from_texts"] = {"value": round(B / el_t, 1), "unit": "chunks/s", "texts": B, "tokens": tm["tokens"],
                                 "frac_of_via_api": round((B / el_t) / (chunks / el_api / world), 4),
                                 "tokenize_s": round(tm["tokenize_s"], 3), "gpu_encode_s": round(tm["encode_s"], 3), "total_s": round(el_t, 3),
                                 "tokenizer_texts_per_s": round(tok_rate, 1), "host_threads": int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)),
                                 "bound": "host tokenizer" if tm["tokenize_s"] > tm["encode_s"] else "gpu",
                                 "tokenizer": "native mi_wordpiece_* (these texts are ASCII; texts it declines go through tokenizers)" if nat else "tokenizers",
                                 "reference_tokenizer_texts_per_s": round(ref_rate, 1), "tokens_equal_reference_tokenizer": tok_equal,
                                 "path": "List[str] -> WordPiece on all host cores -> CSR -> mi_encoder_encode_ragged -> "
                                         "fp32 rows; slabs of 1,024 texts, tokenising overlapped with the GPU",
                                 "finite": bool(np.isfinite(Et).all())}
            # the reranker from STRINGS: 32 queries x 100 passages of ~150 words as (query, passage) pairs through
            # CrossEncoderReranker.predict (native pair tokenizer on all host cores, slabs overlapped with the GPU)
            if "rerank" in res:
                from rag_amd.embeddings import _load_tokenizer
                from rag_amd.retrievers import CrossEncoderReranker, rerank_pairs_csr
                rr = CrossEncoderReranker(encoder=enc, tokenizer=_load_tokenizer(mdir, 512), max_length=512)
                rr.slab_pairs = 800
                qs = synthetic_texts(32, 12, cfg.vocab_size, seed=8)
                ps = synthetic_texts(3200, 150, cfg.vocab_size, seed=9)
                pairs = [(qs[i // 100], ps[i]) for i in range(3200)]
                rr.predict(pairs[:200])
                t0 = time.perf_counter()
                sc = rr.predict(pairs)
                el_rt = time.perf_counter() - t0
                t0 = time.perf_counter()
                f_n, t_n, o_n = rr._pairs_csr(pairs[:800])
                nat_rate = 800 / (time.perf_counter() - t0)
                t0 = time.perf_counter()
                f_r, t_r, o_r = rerank_pairs_csr(pairs[:800], rr.tokenizer, 512)
                lib_rate = 800 / (time.perf_counter() - t0)
                res["rerank"]["from_texts"] = {
                    "value": round(3200 / el_rt, 1), "unit": "pairs/s", "mean_tokens_per_pair": round(float(np.diff(o_n).mean()), 1),
                    "tokenizer": "native mi_wordpiece_encode_pairs" if rr.native_tokenizer is not None else "tokenizers",
                    "tokenizer_pairs_per_s": round(nat_rate, 1), "reference_tokenizer_pairs_per_s": round(lib_rate, 1),
                    "tokens_equal_reference_tokenizer": bool(np.array_equal(f_n, f_r) and np.array_equal(t_n, t_r) and np.array_equal(o_n, o_r)),
                    "finite": bool(np.isfinite(sc).all()),
                    "path": "List[(str, str)] -> pair WordPiece on all host cores -> CSR -> mi_encoder_classify_ragged -> logits; slabs of 800 pairs"}
            del emb
        finally:
            shutil.rmtree(mdir, ignore_errors=True)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        ocfg = oracle.BertConfig(*[getattr(cfg, f) for f, _ in type(cfg)._fields_])
        nb = 2 * oracle.num_threads()
        tc = time.perf_counter()
        Eo = oracle.encode(ocfg, blob, ids[:nb].cpu().numpy(), mask[:nb].cpu().numpy())
        tcpu = time.perf_counter() - tc
        Eg = E[:nb].cpu().numpy()
        cos = (Eg * Eo).sum(1) / (np.linalg.norm(Eg, axis=1) * np.linalg.norm(Eo, axis=1))
        port = {"value": round(nb / tcpu, 3), "unit": "chunks/s", "cores": oracle.num_threads(), "kind": "port",
                "sample": f"oracle.encode (fp32 C port of BertModel + mean-pool) on {nb} of the same chunks in {tcpu:.1f}s"}
        legs = [port]
        res["parity_check"] = {"chunks": nb, "max_1_minus_cos_vs_oracle": float((1 - cos).max())}
        # what sentence-transformers runs (requirements.txt:49 -> transformers BertModel on torch CPU, advanced_rag.py:106
        # "device": "cpu") + its mean-pool / normalise, on the same chunks and weights, all host cores
        try:
            import transformers
            from rag_amd.encoder_weights import unpack_blob
            hc = transformers.BertConfig(vocab_size=cfg.vocab_size, hidden_size=H, num_hidden_layers=nl, num_attention_heads=cfg.num_attention_heads,
                                         intermediate_size=I, max_position_embeddings=cfg.max_position_embeddings,
                                         type_vocab_size=cfg.type_vocab_size, layer_norm_eps=cfg.layer_norm_eps, hidden_act="gelu")
            model = transformers.BertModel(hc, add_pooling_layer=False).eval()
            missing = model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in unpack_blob(cfg, blob).items()}, strict=False)
            assert not missing.unexpected_keys and all("position_ids" in k for k in missing.missing_keys), missing
            nt = 4 * oracle.num_threads()
            ti, tmk = ids[:nt].cpu().long(), mask[:nt].cpu().long()
            with torch.no_grad():
                model(input_ids=ti[:2], attention_mask=tmk[:2])
                tc = time.perf_counter()
                hs = model(input_ids=ti, attention_mask=tmk).last_hidden_state
                mk = tmk.unsqueeze(-1).float()
                Et_ = torch.nn.functional.normalize((hs * mk).sum(1) / mk.sum(1).clamp(min=1e-9), dim=1)
                ttorch = time.perf_counter() - tc
            Eg2 = E[:nt].cpu().numpy()
            cos2 = (Eg2 * Et_.numpy()).sum(1) / np.linalg.norm(Eg2, axis=1)
            legs.append({"value": round(nt / ttorch, 3), "unit": "chunks/s", "cores": torch.get_num_threads(), "kind": "torch-bertmodel",
                         "sample": f"transformers {transformers.__version__} BertModel (fp32, torch CPU) + mean-pool + normalise on {nt} of the same "
                                   f"chunks in {ttorch:.1f}s", "max_1_minus_cos_gpu_vs_bertmodel": float((1 - cos2).max())})
        except Exception as ex:          # noqa: BLE001  (transformers absent or incompatible: the port leg stands alone)
            legs.append({"kind": "torch-bertmodel", "error": repr(ex)[:200]})
        top = max([l for l in legs if "value" in l], key=lambda l: l["value"])
        res["cpu_baseline"] = {"value": top["value"], "unit": "chunks/s", "cores": top["cores"], "kind": top["kind"],
                               "sample": top["sample"], "legs": legs}
    return res


def bench_configs(args, device, dev):
    """The single-GPU BASELINE configs besides the headline, measured in the same driver run:
      config2  1M x 384 fp32 corpus, 1k-query batch, exact top-10, IP on unit vectors (= cosine), MI_MODE_AUTO
      config5  50M x 768 bf16 corpus, 10k-query batch, exact top-100 (the list a cross-encoder rerank batch is cut from)
    wall = the whole mi_index_search_dev call (prep + scan + finish + certify + read-back of the certificate count),
    scan = the dominant kernel alone (HIP events inside the library)."""
    from rag_amd.index import FlatIndex
    out = {}

    def queries(n, d):
        g = FlatIndex(d, metric="ip", dtype="f32", device=device)
        g.add_synthetic(n, seed=1, normalize=True)
        q = torch.from_numpy(g.reconstruct_n(0, n)).to(dev)
        del g
        return q

    def run(name, n, d, dtype, nq, k, warm, steps, extra):
        idx = FlatIndex(d, metric="ip", dtype=dtype, device=device)
        idx.reserve(n)
        idx.add_synthetic(n, seed=0, normalize=True)
        q = queries(nq, d)
        for _ in range(warm):
            idx.search_torch(q, k)
        st0 = idx.stats()
        wall, scan = [], []
        for _ in range(steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            idx.search_torch(q, k)
            torch.cuda.synchronize()
            wall.append((time.perf_counter() - t0) * 1e3)
            scan.append(idx.last_scan_time()["ms"])
        st1 = idx.stats()
        w, sc = float(np.median(wall)), float(np.median(scan))
        fl = 2.0 * nq * n * d
        by = float(n) * d * (2 if dtype == "bf16" else 4)
        res = {"workload": f"exact top-{k}, {n}x{d} {dtype} corpus, {nq}-query batch, IP on unit vectors (cosine), one GPU",
               "wall_ms": round(w, 3), "scan_ms": round(sc, 3), "wall_over_scan": round(w / sc, 3), "queries_per_s": round(nq / (w * 1e-3), 1),
               "scan_kernel": idx.last_scan_time()["kind"],
               "roofline": {"bound": "mfma", "achieved": round(fl / (sc * 1e-3) / 1e12, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(fl / (sc * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                            "floor_ms": {"hbm": round(by / (HBM_PEAK_GBS * 1e9) * 1e3, 3), "mfma": round(fl / (MFMA_BF16_PEAK_TFLOPS * 1e12) * 1e3, 3)},
                            "traffic": pmc_traffic("scan", rows_per_gpu=n, dim=d, dtype=dtype, queries=nq)},
               "queries_certified": st1["fast"] - st0["fast"], "queries_exact_rerun": st1["exact"] - st0["exact"], "steps": steps}
        res.update(extra)
        out[name] = res
        del idx, q
        torch.cuda.empty_cache()

    run("config2", 1_000_000, 384, "f32", 1000, 10, 3, 7,
        {"note": "fp32 index: the scan filters on a bf16 shadow of the rows, scores are the canonical fp32 chain of the fp32 rows"})
    run("config5", 50_000_000, 768, "bf16", 10_000, 100, 1, 2, {})
    return out


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=int, default=100_000_000, help="total corpus rows over all ranks")
    ap.add_argument("--dim", type=int, default=384)
    ap.add_argument("--queries", type=int, default=10_000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=2_000_000)
    ap.add_argument("--cpu-queries", type=int, default=4096)
    ap.add_argument("--blas-rows", type=int, default=500_000, help="rows of the sgemm + argpartition CPU leg")
    ap.add_argument("--blas-queries", type=int, default=1024)
    ap.add_argument("--check", type=int, default=8, help="queries verified against the oracle on the CPU sample rows")
    ap.add_argument("--embed-chunks", type=int, default=4096, help="256-token chunks per rank per embed step (0 = skip)")
    ap.add_argument("--embed-steps", type=int, default=3)
    ap.add_argument("--embed-only", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (BASELINE configs 2 and 5 on this GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    ngpu = world
    # Rehearsal switches (a one-GPU box cannot run RCCL with two ranks on one device): MI355RAG_BENCH_BACKEND=gloo carries the
    # collectives, MI355RAG_BENCH_ONE_GPU=1 puts every rank on device 0.  The driver's runs use neither.
    backend = os.environ.get("MI355RAG_BENCH_BACKEND", "nccl")
    if os.environ.get("MI355RAG_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from rag_amd.distributed import ShardedFlatIndex
    from rag_amd.index import FlatIndex

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.embed_only:
        res = bench_embed(args, dev, rank, world, barrier)
        if rank == 0:
            print(json.dumps(res), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return 0

    N, d, Q, k = args.rows, args.dim, args.queries, args.k
    shard = (N + ngpu - 1) // ngpu
    row0 = rank * shard
    nloc = max(0, min(N, row0 + shard) - row0)

    t0 = time.time()
    idx = FlatIndex(d, metric="ip", dtype=args.dtype, device=local_rank)
    idx.set_id_base(row0)
    idx.reserve(nloc)
    idx.add_synthetic(nloc, seed=0, normalize=True)
    # queries: same generator, another seed, produced on device and kept there
    qgen = FlatIndex(d, metric="ip", dtype="f32", device=local_rank)
    qgen.add_synthetic(Q, seed=1, normalize=True)
    q_host = qgen.reconstruct_n(0, Q)
    del qgen
    q_dev = torch.from_numpy(q_host).to(dev)
    torch.cuda.synchronize()
    log(f"[rank {rank}] corpus shard {nloc} x {d} {args.dtype} + {Q} queries ready in {time.time() - t0:.1f}s")

    scan_ms, scan_launches = [], []

    # rank-local search into the rank's packed block, ONE all-gather of the blocks, merge in place
    sharded = ShardedFlatIndex(d, metric="ip", dtype=args.dtype, device=local_rank, local_index=idx) if world > 1 else None

    def step():
        if world > 1:
            return sharded.search_torch(q_dev, k)
        return idx.search_torch(q_dev, k)

    for _ in range(args.warmup):
        step()
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        D, I = step()
        st = idx.last_scan_time()
        scan_ms.append(st["ms"])
        scan_launches.append(st["launches"])
    barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stats = idx.stats()
    kind = idx.last_scan_time()["kind"]

    # ---- the HBM-bound regime of the same scan (intensity = Q flop/B, so the scan is
    # HBM-bound below Q ~ 300): the app's own Q = 1 and one full query tile, on the same resident shard
    regimes = []
    for Qs in (1, 256):
        if Qs > Q:
            continue
        qs = q_dev[:Qs].contiguous()
        idx.search_torch(qs, k)
        ms = []
        for _ in range(3):
            idx.search_torch(qs, k)
            ms.append(idx.last_scan_time()["ms"])
        m = float(np.median(ms))
        gbs = float(nloc) * d * (2 if args.dtype == "bf16" else 4) / (m * 1e-3) / 1e9
        regimes.append({"queries": Qs, "scan_ms": round(m, 3), "achieved_GBs": round(gbs, 1),
                        "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4), "qps_per_gpu": round(Qs / (m * 1e-3), 1),
                        "algorithmic_GB": round(float(nloc) * d * (2 if args.dtype == "bf16" else 4) / 1e9, 2),
                        "traffic_GB": pmc_traffic("scan", rows_per_gpu=nloc, dim=d, dtype=args.dtype, queries=Qs)})
    # ---- CPU sample rows (read back before the big index is released) and the live parity check: the CPU leg below computes
    # oracle.search on the first `crow` rows x `cq` queries; a second index built from the same generator holds exactly those
    # rows, the GPU answers the same queries on it and the two are compared bit for bit (ids and score bits).
    cpu_rows = None
    check_gpu = None
    if rank == 0 and ngpu == 1 and not args.no_cpu_baseline:
        crow = min(args.cpu_rows, nloc)
        cq = min(args.cpu_queries, Q)
        cpu_rows = idx.reconstruct_n(0, crow)            # the stored (bf16-rounded) values, as fp32
        if args.check > 0:
            sub = FlatIndex(d, metric="ip", dtype=args.dtype, device=local_rank)
            sub.add_synthetic(crow, seed=0, normalize=True)
            Ds, Is = sub.search_torch(q_dev[:cq].contiguous(), k)
            check_gpu = (Ds.cpu().numpy(), Is.cpu().numpy(), sub.last_scan_time()["kind"], sub.stats())
            del sub, Ds, Is

    # ---- the other single-GPU BASELINE configs on the same device, after the headline index is released
    configs = None
    if rank == 0 and ngpu == 1 and not args.no_configs and (N, d, k, args.dtype) == (100_000_000, 384, 10, "bf16"):
        del idx, D, I
        torch.cuda.empty_cache()
        configs = bench_configs(args, local_rank, dev)
        idx = None

    embed = None
    if args.embed_chunks > 0:
        embed = bench_embed(args, dev, rank, world, barrier)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return 0

    ms_per_step = elapsed / args.steps * 1e3
    value = Q * args.steps / elapsed

    # ---- roofline of the dominant kernel (the scan), measured with HIP events inside the library
    launches = max(1, int(np.sum(scan_launches)))
    avg_ms = float(np.sum(scan_ms)) / launches
    flops_per_launch = 2.0 * Q * nloc * d * args.steps / launches      # algorithmic: 2*Q*N_local*d per pass
    bytes_per_launch = float(nloc) * d * (2 if args.dtype == "bf16" else 4) * args.steps / launches
    if kind == "mfma_bf16":
        ach = flops_per_launch / (avg_ms * 1e-3) / 1e12
        roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": pmc_traffic("scan", rows_per_gpu=nloc, dim=d, dtype=args.dtype, queries=Q),
                    "traffic_unit": "GB per launch (FETCH_SIZE PMC pass, profiles/scan_traffic.json; null when the kernel source changed since)",
                    "kernel": "mfma_scan_kernel", "avg_launch_ms": round(avg_ms, 3),
                    "algorithmic_hbm_GBs": round(bytes_per_launch / (avg_ms * 1e-3) / 1e9, 1)}
    else:
        ach = flops_per_launch / (avg_ms * 1e-3) / 1e12
        roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": 157.3, "unit": "TFLOP/s",
                    "frac": round(ach / 157.3, 4), "traffic": None, "kernel": "scan_dense_kernel",
                    "avg_launch_ms": round(avg_ms, 3)}

    # ---- CPU baseline, bounded samples of the same workload, two legs:
    #   port       the oracle: AVX2 fmaf-chain restatement of IndexFlat::search (bit-exact with the GPU path)
    #   port-blas  what faiss does for nq >= 20 and what SURVEY.md section 8d planned: blocked fp32 sgemm (NumPy/BLAS)
    #              + argpartition + sort of the k winners; its summation order is the BLAS's, so it is timed, not compared
    cpu = None
    check = None
    if not args.no_cpu_baseline and ngpu == 1:
        import oracle
        rows = cpu_rows
        odt = oracle.DTYPE_BF16 if args.dtype == "bf16" else oracle.DTYPE_F32
        rows_in = oracle.round_bf16(rows) if args.dtype == "bf16" else rows
        tc = time.perf_counter()
        Do, Io = oracle.search(rows_in, q_host[:cq], k, metric=oracle.METRIC_IP, dtype=odt)
        tcpu = time.perf_counter() - tc
        # extrapolate linearly in N to the full corpus
        port = {"value": cq / tcpu * (crow / float(N)), "unit": "queries/s", "cores": oracle.num_threads(), "kind": "port",
                "sample": f"oracle.search (AVX2 fmaf-chain port of IndexFlat) on the first {crow} rows x {cq} queries "
                          f"in {tcpu:.2f}s, extrapolated linearly in N to {N} rows"}
        if check_gpu is not None:
            Dg, Ig, gkind, gst = check_gpu
            ok = bool(np.array_equal(Ig, Io) and np.array_equal(Dg.view(np.uint32), Do.view(np.uint32)))
            check = {"queries": cq, "rows": crow, "k": k, "bit_exact_vs_oracle": ok, "gpu_scan_kernel": gkind,
                     "queries_certified": gst["fast"], "queries_exact_rerun": gst["exact"],
                     "what": "ids and score bits of the GPU answer on an index of the corpus' first rows (same generator) against "
                             "oracle.search on those rows: the CPU leg's own result, so the check costs one small GPU search"}
        brow, bq = min(args.blas_rows, crow), min(args.blas_queries, Q)
        qb = np.ascontiguousarray(q_host[:bq])
        if args.dtype == "bf16":
            qb = oracle.bf16_to_f32(oracle.round_bf16(qb))
        tc = time.perf_counter()
        best_s = np.full((bq, k), -np.inf, dtype=np.float32)
        best_i = np.full((bq, k), -1, dtype=np.int64)
        blk = 65536
        for r0 in range(0, brow, blk):
            S = qb @ rows[r0:r0 + blk].T                                      # sgemm, [bq][blk]
            kk = min(k, S.shape[1])
            part = np.argpartition(-S, kk - 1, axis=1)[:, :kk]
            cand_s = np.concatenate([best_s, np.take_along_axis(S, part, axis=1)], axis=1)
            cand_i = np.concatenate([best_i, part + r0], axis=1)
            o = np.lexsort((cand_i, -cand_s), axis=1)[:, :k]
            best_s, best_i = np.take_along_axis(cand_s, o, axis=1), np.take_along_axis(cand_i, o, axis=1)
        tblas = time.perf_counter() - tc
        try:
            from threadpoolctl import threadpool_info
            bthreads = max([p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"] or [1])
        except Exception:
            bthreads = int(os.environ.get("OMP_NUM_THREADS", "1"))
        agree = float(np.mean(best_i[:min(bq, cq)] == Io[:min(bq, cq)])) if brow == crow else None
        blas = {"value": bq / tblas * (brow / float(N)), "unit": "queries/s", "cores": bthreads, "kind": "port-blas",
                "sample": f"NumPy sgemm + argpartition + sort on the first {brow} rows x {bq} queries in {tblas:.2f}s "
                          f"(blocks of {blk} rows), extrapolated linearly in N to {N} rows",
                "ids_equal_to_port": agree}
        top = blas if blas["value"] > port["value"] else port
        cpu = {"value": top["value"], "unit": "queries/s", "cores": top["cores"], "kind": top["kind"],
               "sample": top["sample"], "legs": [port, blas]}

    out = {
        "metric": ("queries/sec exact top-10 @100Mx384 bf16" if (N, d, k, args.dtype) == (100_000_000, 384, 10, "bf16")
                   else f"queries/sec exact top-{k} @{N}x{d} {args.dtype}"),
        "value": round(value, 2),
        "unit": "queries/s",
        "n_gpus": ngpu,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": f"exact top-{k}, {N}x{d} {args.dtype} corpus (synthetic unit vectors, seed 0), "
                               f"{Q}-query batch, IP=cosine, row-sharded over {ngpu} GPU(s)",
                   "rows": N, "dim": d, "queries": Q, "k": k, "parallelism": f"row-shard x{ngpu} + all-gather merge"},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "path": {"scan_kernel": kind, "queries_certified": stats["fast"], "queries_exact_rerun": stats["exact"]},
    }
    out["hbm_bound_regime"] = {"note": "same kernel, same resident shard, small query batches; peak 8000 GB/s (spec), "
                                       "~6300 GB/s measured streaming ceiling (MI355X_MICROARCH.md)", "cases": regimes}
    if check is not None:
        out["parity_check"] = check
    if configs is not None:
        out["configs"] = configs
    if embed is not None:
        out["embed"] = embed
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
