#!/usr/bin/env python3
"""Headline benchmark: greedy bs=1 decode tokens/s of a random-init Llama-2-7B whose linear
layers are E8P12 2-bit QuantLinear (BASELINE.json configs[1]), on N GPUs of one node.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

* a "step" is one decoded token = one pass of the hot path (224 QuantLinear forwards, plus the
  attention / norm / lm_head glue) replayed from a captured hipGraph;
* multi-GPU = independent replicas (the Hadamard transform precludes tensor parallelism, SURVEY
  8e): every rank decodes its own sequence, no data-path collective; value = all ranks' tokens /
  max-over-ranks time; scaling "weak";
* `roofline`: the dominant kernel.  Llama-2-7B E8P12 decodes a token with ONE persistent launch for all 32 blocks
  (`decode_block_kernel`, csrc/decode_block.hip): it is timed live with HIP events on the launch stream (graph replay of
  that launch alone), achieved = the algorithmic bytes of the 224 QuantLinear calls it contains (codes + x + y + SU / SV,
  SURVEY 8d) / its duration; peak = 8 TB/s.  Models the persistent launch does not take (70B, other codebooks) report
  the stage-wise step's GEMV launches the same way (4 per block, each on its own layer's weights).  `traffic` = measured
  HBM bytes per launch from the PMC pass committed under profiles/ (null when that file is absent);
* `cpu_baseline`: the CPU oracle (a port of the reference semantics: the reference has no CPU
  inference path) timed on this host for one layer of each Llama-7B shape and extrapolated to
  tokens/s.
Prints ONE JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

METRIC = "decode tokens/sec bs=1 Llama-7B E8P12 2-bit, 1xMI355X; % HBM roofline"
HBM_PEAK_GBPS = 8000.0


def qidxs_nbytes(m):
    """bytes of a module's code matrix (decode.qidxs_nbytes: the row-major tensor, or the tiled copy that replaced it)"""
    from quip_for_all_amd.decode import qidxs_nbytes as f
    return f(m)


def gemv_roofline(dec):
    """live HIP-event timing of the GEMV launches of one decode step exactly as the decoder issues
    them (per block: q/k/v group, o, gate/up group, down; each on its own layer's weights), bs=1
    planes path, replayed from a hipGraph on the launch stream"""
    import quip_for_all_amd  # noqa: F401
    from quip_for_all_amd import capi
    L = capi.lib()
    dev = dec.dev
    from quip_for_all_amd.qlinear import _GROUP_MAX_BYTES, _gemv_planes_grouped
    calls = []     # (modules, [planes])
    algo = 0
    launches = 0
    for layer in dec.layers:
        for names in (("q", "k", "v"), ("o",), ("gate", "up"), ("down",)):
            ms = [layer[n] for n in names]
            k = ms[0].q_in_features
            planes = []
            for m in ms:
                x = torch.randn(1, k, device=dev).half()
                pl = torch.empty(L.quip_e8p_planes_bytes(k), dtype=torch.uint8, device=dev)
                capi.check(L.quip_e8p_x_to_planes(x.data_ptr(), pl.data_ptr(), k,
                                                  torch.cuda.current_stream().cuda_stream), "x_to_planes")
                planes.append(pl)
                algo += m.q_out_features * k // 4 + 2 * k + 2 * m.q_out_features
            calls.append((ms, planes))
            nbytes = sum(qidxs_nbytes(m) for m in ms)
            launches += 1 if (len(ms) == 1 or nbytes <= _GROUP_MAX_BYTES) else len(ms)   # the decoder's grouping policy

    def run():
        for ms, planes in calls:
            _gemv_planes_grouped(ms, planes)
    run()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.cuda.graph(graph, stream=side):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        graph.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    t = float(np.median(ts[1:]))
    per_launch_bytes = algo / launches
    per_launch_s = t / launches
    achieved = per_launch_bytes / per_launch_s / 1e9
    # measured HBM bytes per GEMV launch: the committed PMC pass of THIS workload (tools/prof_bench.sh),
    # one file per model; null when there is none for the model being run
    traffic = traffic_src = None
    llama2 = dec.s.ffn in (11008, 28672)     # (the committed passes are of the Llama-2 shapes: another ffn / KV width is another workload)
    name = "gemv_hbm_traffic.json" if dec.s.hidden == 4096 and dec.s.layers == 32 and llama2 else \
        f"gemv_hbm_traffic_h{dec.s.hidden}_l{dec.s.layers}.json" if llama2 else \
        f"gemv_hbm_traffic_h{dec.s.hidden}_f{dec.s.ffn}_l{dec.s.layers}.json"
    pf = os.path.join(REPO, "profiles", name)
    if os.path.exists(pf):
        try:
            j = json.load(open(pf))
            traffic = j.get("hbm_bytes_per_launch")
            # PMC counters need rocprofv3 around the process, so this figure is not measured in this run: it is the
            # committed FETCH_SIZE pass of the same workload (tools/prof_bench.sh), stamped with where it came from
            traffic_src = "profiles/%s (rocprofv3 --pmc FETCH_SIZE x2, %s)" % (name, j.get("measured_at", "round 1"))
        except Exception:
            traffic = None
    big = any(qidxs_nbytes(m) >= (16 << 20) and m.q_in_features >= 8192 for layer in dec.layers[:1]
              for m in layer.values() if hasattr(m, "Qidxs"))
    return {"bound": "hbm", "kernel": "e8p_gemv_v2_kernel (launches >= 16 MB at k >= 8192) / e8p_gemv_mfma_kernel" if big
            else "e8p_gemv_mfma_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "launches": launches, "algorithmic_bytes_per_launch": round(per_launch_bytes),
            "mean_launch_us": round(per_launch_s * 1e6, 3)}


def engine_roofline(dec):
    """the persistent token launch (decode_block_kernel) alone, replayed from a hipGraph and timed with HIP events on
    the launch stream; algorithmic bytes = SURVEY 8d's per-call figure summed over the launch's QuantLinear calls"""
    import math
    dev = dec.dev
    s = dec.s
    h = dec.embed[:1].reshape(-1).clone()
    pos = torch.full((1,), 64, dtype=torch.long, device=dev)
    args = (dec.eng_layers, h, pos, dec.cos, dec.sin, dec.eng_grid, dec.eng_ws, len(dec.layers), dec.max_len, s.rms_eps,
            1.0 / math.sqrt(s.head_dim), None, -1, dec.eng_codebook, dec.eng_resid_scale, getattr(dec, "eng_shape", 0))
    torch.ops.quip_lib.block_engine(*args)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.cuda.graph(graph, stream=side):
        torch.ops.quip_lib.block_engine(*args)
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        graph.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    t = float(np.median(ts[2:]))
    algo = 0
    for layer in dec.layers:
        for m in layer.values():
            if hasattr(m, "Qidxs"):
                algo += qidxs_nbytes(m) + 4 * (m.in_features + m.out_features)
    achieved = algo / t / 1e9
    traffic = traffic_src = None
    gqa = getattr(dec, "eng_shape", 0) == 1
    # the committed PMC pass of THIS workload: one file per model shape (7B: engine_hbm_traffic.json)
    fname = "engine_hbm_traffic.json" if (s.hidden == 4096 and len(dec.layers) == 32 and s.ffn == 11008) else \
        "engine_hbm_traffic_h%d_f%d_l%d.json" % (s.hidden, s.ffn, len(dec.layers)) if s.hidden == 4096 else \
        "engine_hbm_traffic_h%d_l%d.json" % (s.hidden, len(dec.layers))
    pf = os.path.join(REPO, "profiles", fname)
    if os.path.exists(pf):
        try:
            j = json.load(open(pf))
            traffic = j.get("hbm_bytes_per_launch")
            traffic_src = "profiles/%s (rocprofv3 --pmc FETCH_SIZE x2, %s)" % (fname, j.get("measured_at", "?"))
        except Exception:
            traffic = None
    return {"bound": "hbm", "kernel": "%s (one persistent launch per token: all %d blocks)" % (
                "decode_block_gqa_kernel" if gqa else ("decode_block_kernel (decode_block_g8.hip)" if getattr(dec, "eng_shape", 0) == 2
                                                       else "decode_block_kernel"), len(dec.layers)),
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
            # (`traffic`: NOT measured beside this timed region -- read from the committed rocprofv3 --pmc pass of the same launch
            #  on another box, named in traffic_source)
            "traffic": traffic, "traffic_is_from_file": traffic is not None, "traffic_source": traffic_src, "launches": 1,
            "algorithmic_bytes_per_launch": algo,
            "mean_launch_us": round(t * 1e6, 1), "us_per_block": round(t * 1e6 / len(dec.layers), 2),
            "engine_status": dec.engine_status()}


def gqa_phase_rates(dec):
    """The E8P12 products at the 70B layer shapes as they run INSIDE the persistent launch (decode_block_gqa_kernel): clock
    stamps (s_memtime, workgroup mean) around the product phases of one block in the middle of the model, converted to time
    with the same launch's own clocks-per-microsecond (block span in clocks / HIP-event time per block).  A phase = the 8
    waves of every workgroup multiplying their items of the matrices named, weights streamed through the launch-long ring
    (nine 2 KB requests ahead per wave), accumulation and -- where it is part of the phase -- the 16-granule publication;
    bytes = the matrices' code bytes.  `GBps` = code bytes MULTIPLIED per second inside the phase: the rate of the
    decode GEMV at these shapes with its prologue (tables, planes) and tail amortised over the launch.  It is not the HBM
    rate of the phase: the ring had requested the first nine items of every wave before the phase began, so only
    (items - 9) / items of the bytes had to LAND inside it -- `GBps_landed_at_least` is that lower bound on the HBM side."""
    import math
    s = dec.s
    L = len(dec.layers)
    dl = L // 2
    h = dec.embed[:1].reshape(-1).clone()
    pos = torch.full((1,), 40, dtype=torch.long, device=dec.dev)
    dbg = torch.zeros(256 * 32, dtype=torch.int64, device=dec.dev)
    args = (dec.eng_layers, h, pos, dec.cos, dec.sin, dec.eng_grid, dec.eng_ws, L, dec.max_len, s.rms_eps,
            1.0 / math.sqrt(s.head_dim), dbg, dl, 0, 0.0, 1)
    runs = []
    for it in range(6):
        dbg.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        torch.ops.quip_lib.block_engine(*args)
        b.record()
        torch.cuda.synchronize()
        if it >= 2:
            runs.append((dbg.cpu().numpy().reshape(256, 32).astype(np.float64), a.elapsed_time(b) * 1e3 / L))
    hid, ffn, kvw = s.hidden, s.ffn, s.kv_heads * s.head_dim
    # (stamp before, stamp after, code bytes, items per wave)
    phases = {"gate_up (2 x %dx%d)" % (ffn, hid): (27, 11, 2 * ffn * hid // 4, 28),
              "down (%dx%d) + publish" % (hid, ffn): (15, 16, hid * ffn // 4, 14),
              "o (%dx%d) + publish" % (hid, hid): (8, 9, hid * hid // 4, 4),
              "q, k, v (%dx%d + 2 x %dx%d) + publish" % (hid, hid, kvw, hid): (22, 3, (hid + 2 * kvw) * hid // 4, 6)}
    out = {}
    tpu = float(np.median([(d[:, 17] - d[:, 0]).mean() / us for d, us in runs]))        # clocks per microsecond (block span / block time)
    for name, (s0, s1, nbytes, items) in phases.items():
        ticks = float(np.median([(d[:, s1] - d[:, s0]).mean() for d, _ in runs]))
        us = ticks / tpu
        out[name] = {"code_bytes": nbytes, "clocks": round(ticks), "us": round(us, 2), "GBps": round(nbytes / us / 1e3, 1),
                     "frac_of_8TBps": round(nbytes / us / 1e3 / HBM_PEAK_GBPS, 4), "items_per_wave": items,
                     "GBps_landed_at_least": round(nbytes * max(items - 9, 0) / items / us / 1e3, 1)}
    out["clocks_per_us"] = round(tpu, 1)
    out["block"] = dl
    out["engine_status"] = dec.engine_status()
    return out


def gqa_stream_rate(dec, launches=12, warm=12):
    """HBM-landed rate of the E8P12 decode GEMV at the 70B layer shapes INSIDE the persistent launch, measured:
    the launch in its measurement mode (decode_block_gqa.hip, dbg_layer = -2) runs the products of all blocks -- the same 54
    items per wave and block through the same ring, decode and MFMAs: q k v o gate up down of every block -- and leaves out the
    edges, the attention and the hand-offs, so the weight stream never waits for an input.  Every byte multiplied in the timed
    region was requested AND landed in it (no pre-filled ring: the first nine items of a wave against 54 x layers): GBps = all
    code bytes of the model / HIP-event time of the launch.  The rocprofv3 FETCH_SIZE of the same launch is in profiles/
    (tools/prof_gqa_stream.sh).  The products' results are not used (round 6: the launch fills its digit planes with pseudo-random
    digits in this mode, so that the matrix cores switch as in a real launch -- the rate is a power figure too: the launch runs
    at the clock its power allows).  `warm` untimed launches first (~40 ms): the clock of an idle device ramps over the first
    launches of a run (profiles/r06_gqa_stream.txt: 1.55 -> 1.95 GHz over eight), and the figure asked for is the sustained
    one; every timed launch is listed."""
    import math
    s = dec.s
    L = len(dec.layers)
    h = dec.embed[:1].reshape(-1).clone()
    pos = torch.full((1,), 40, dtype=torch.long, device=dec.dev)
    args = (dec.eng_layers, h, pos, dec.cos, dec.sin, dec.eng_grid, dec.eng_ws, L, dec.max_len, s.rms_eps,
            1.0 / math.sqrt(s.head_dim), None, -2, 0, 0.0, 1)
    ts = []
    for it in range(launches + warm):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        torch.ops.quip_lib.block_engine(*args)
        b.record()
        torch.cuda.synchronize()
        if it >= warm:
            ts.append(a.elapsed_time(b) * 1e3)
    us = float(np.median(ts))
    code_bytes = sum(qidxs_nbytes(L_[k]) for L_ in dec.layers for k in ("q", "k", "v", "o", "gate", "up", "down"))
    st = dec.engine_status()
    dec.engine_reset()                  # (the mode publishes nothing: leave the workspace as a fresh one)
    return {"mode": "products of all %d blocks in one launch, no edges / hand-offs (decode_block_gqa.hip, dbg_layer = -2)" % L,
            "code_bytes": code_bytes, "us_per_launch": round(us, 1), "us_per_block": round(us / L, 2),
            "GBps": round(code_bytes / us / 1e3, 1), "frac_of_8TBps": round(code_bytes / us / 1e3 / HBM_PEAK_GBPS, 4),
            "launch_us_min_max": [round(min(ts), 1), round(max(ts), 1)], "launches_us": [round(t, 1) for t in ts],
            "warm_launches": warm, "engine_status": st}


def gemv_per_shape(shapes, dev, pool_bytes=640 << 20, iters=100, warm_ms=40.0):
    """SURVEY 8d layer micro-bench inside the bench run: the default bs=1 E8P12 GEMV entry point on every (n, k) of
    `shapes`, weights cycled through a pool larger than the 256 MB Infinity Cache, `iters` launches per graph replay,
    HIP-event timed; frac = algorithmic bytes (codes + x + y) / time / 8 TB/s.  The timed replays follow `warm_ms` of untimed ones:
    an idle device starts at ~1.5 GHz and takes tens of milliseconds of load to reach its sustained clock (the same ramp
    gqa_stream_rate's warm launches are for: profiles/r06_gqa_stream.txt); the first replay's figure is reported beside the
    sustained one"""
    import quip_for_all_amd as Q
    from quip_for_all_amd import capi
    L = capi.lib()
    grid = Q.codebook.codebook_id["E8P12"](inference=True).to(dev).grid_packed_abs
    st = lambda: torch.cuda.current_stream().cuda_stream   # noqa: E731
    out = {}
    import ctypes
    for shape in shapes:
        # (n, k): one matrix through quip_e8p_gemv_planes_ws; ((n0, n1, ..), k): the modules that read the same activation (q / k / v,
        # gate / up) in ONE launch through quip_e8p_gemv_planes_group_ws -- what QuantLinear's grouped forward calls
        ns, k = (shape[0], shape[1]) if isinstance(shape[0], tuple) else ((shape[0],), shape[1])
        n = sum(ns)
        cnt = len(ns)
        wbytes = n * k // 4
        npool = max(4, pool_bytes // wbytes + 1)
        g = torch.Generator(device=dev).manual_seed(0)
        pool = [[torch.randint(-32768, 32767, (m, k // 8), generator=g, dtype=torch.int32, device=dev).to(torch.int16) for m in ns]
                for _ in range(npool)]
        x = torch.randn(1, k, device=dev).half()
        planes = torch.empty(L.quip_e8p_planes_bytes(k), dtype=torch.uint8, device=dev)
        capi.check(L.quip_e8p_x_to_planes(x.data_ptr(), planes.data_ptr(), k, st()), "planes")
        ys = [torch.empty(1, m, dtype=torch.float16, device=dev) for m in ns]
        ws = torch.zeros(max(sum(L.quip_e8p_gemv_workspace_bytes(m) for m in ns) // 4, 1), dtype=torch.int32, device=dev)
        if cnt == 1:
            call = lambda i: capi.check(L.quip_e8p_gemv_planes_ws(planes.data_ptr(), pool[i % npool][0].data_ptr(), grid.data_ptr(),   # noqa: E731
                                                                  ys[0].data_ptr(), n, k, ws.data_ptr(), ws.numel() * 4, st()), "gemv")
        else:
            vp = ctypes.c_void_p * cnt
            pl, yp, nsa = vp(*[planes.data_ptr()] * cnt), vp(*[y.data_ptr() for y in ys]), (ctypes.c_int32 * cnt)(*ns)
            qps = [vp(*[q.data_ptr() for q in qs]) for qs in pool]
            call = lambda i: capi.check(L.quip_e8p_gemv_planes_group_ws(pl, qps[i % npool], grid.data_ptr(), yp, nsa, cnt, k,   # noqa: E731
                                                                        ws.data_ptr(), ws.numel() * 4, st()), "gemv group")
        for i in range(3):
            call(i)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.cuda.graph(gr, stream=side):
            for i in range(iters):
                call(i)
        torch.cuda.synchronize()
        def timed():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            gr.replay()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) * 1e3 / iters
        cold = timed()                                  # (the first replay: the clock an idle device starts with)
        t0 = time.perf_counter()
        warm = 0
        while (time.perf_counter() - t0) * 1e3 < warm_ms:
            gr.replay()
            warm += 1
        torch.cuda.synchronize()
        ts = [timed() for _ in range(5)]
        us = sorted(ts)[2]
        algo = n * k // 4 + 2 * k + 2 * n
        out[("%dx%d" % (n, k)) if cnt == 1 else ("(%s)x%d, one launch" % (" + ".join(str(m) for m in ns), k))] = {"algorithmic_bytes": algo, "us_per_launch": round(us, 2), "GBps": round(algo / us / 1e3, 1),
                                 "frac": round(algo / us / 1e3 / HBM_PEAK_GBPS, 4), "us_min_max": [round(min(ts), 2), round(max(ts), 2)],
                                 "us_first_replay": round(cold, 2), "warm_replays": warm}
        del pool, gr, call
        torch.cuda.empty_cache()
    return out


def cpu_baseline(budget_s=25.0):
    """CPU oracle ("port": the reference has no CPU path) on one layer of each 7B shape; tokens/s =
    1 / (32 * (4 t_4096x4096 + 2 t_11008x4096 + t_4096x11008) + lm_head)."""
    from oracle import c_oracle
    shapes = {"attn 4096->4096": (4096, 4096, 4), "gate/up 4096->11008": (4096, 11008, 2),
              "down 11008->4096": (11008, 4096, 1)}
    threads = c_oracle.num_threads()
    per_layer = 0.0
    detail = {}
    t_start = time.time()
    for name, (fin, fout, mult) in shapes.items():
        t = c_oracle.time_qlinear_forward("E8P12", fin, fout, min_time=budget_s / 6.0)
        detail[name] = round(t * 1e3, 3)
        per_layer += mult * t
    lm = c_oracle.time_dense_gemv(32000, 4096, min_time=budget_s / 12.0)
    tok_s = 1.0 / (32 * per_layer + lm)
    return {"value": round(tok_s, 4), "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": "C oracle (decode + butterfly Hadamard + fp32 GEMV, OpenMP x%d) timed on one QuantLinear of "
                      "each Llama-2-7B shape %s ms + fp16 lm_head %.2f ms, extrapolated to a 32-layer token; "
                      "%.0f s of CPU work" % (threads, json.dumps(detail), lm * 1e3, time.time() - t_start)}


def prefill_config5(dec, batch=16, seq=2048):
    """BASELINE.json configs[4] (bs=16 x seq=2048 prefill, the MFMA batched-GEMM path) as an extra of the N=1 line:
    the seven QuantLinear forwards of ONE decoder block of the benchmarked model on M = batch * seq rows, HIP-event timed on
    BOTH batched paths: `decompress_gemm` (the default: own batch Hadamard kernels + own decompress launch + the VENDOR fp16
    GEMM -- the reference's shape, e8p12.py:151-155) and `fused` (own batch Hadamard kernels + the hand-written fused dequant +
    MFMA tile kernel, csrc/e8p_prefill_gemm.hip: no dense W, no vendor library).  The top-level figures are the default's;
    MFMA roofline = 2 M in out flops over the 2.5 PFLOP/s dense fp16 peak (MI355X_MICROARCH.md)."""
    import torch
    L = dec.layers[0]
    M = batch * seq
    dev = dec.dev
    mods = [L[k] for k in ("q", "k", "v", "o", "gate", "up", "down")]
    xs = {m.in_features: torch.randn(M, m.in_features, device=dev, dtype=torch.float16) for m in mods}
    flops = sum(2.0 * M * m.in_features * m.out_features for m in mods)
    from quip_for_all_amd.codebook.codebooks import E8P12_codebook

    def time_mode(mode):
        saved_mode = E8P12_codebook.batched_mode
        E8P12_codebook.batched_mode = mode
        try:
            with torch.no_grad():
                for m in mods:      # warm-up (allocator, GEMM heuristics)
                    m(xs[m.in_features])
                torch.cuda.synchronize()
                ts = []
                for _ in range(3):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for m in mods:
                        m(xs[m.in_features])
                    b.record()
                    torch.cuda.synchronize()
                    ts.append(a.elapsed_time(b))
            regime = mods[0].codebook.batched_regime(M, mods[0].q_out_features, mods[0].q_in_features)
        finally:
            E8P12_codebook.batched_mode = saved_mode
        t = sorted(ts)[1]
        return {"gemm_path": regime, "ms_per_block": round(t, 3), "tflops": round(flops / t / 1e9, 1),
                "frac_of_2500_tflops": round(flops / t / 1e9 / 2500.0, 4)}
    # BOTH batched paths, every time: the default (own decompress launch + the vendor fp16 GEMM, the
    # reference's shape, e8p12.py:151-155) and the hand-written fused dequant + MFMA tile kernel (QUIP_BATCHED_MM=fused)
    both = {"decompress_gemm": time_mode("auto"), "fused": time_mode("fused")}
    ms = both["decompress_gemm"]["ms_per_block"]
    # time to first token of ONE 2048-token prompt through the whole model (LlamaDecoder.prefill: all blocks incl.
    # causal attention, rotary embedding, cache fill, final norm + lm_head of the last token)
    ttft = None
    try:
        import quip_for_all_amd.decode as Dm
        d2 = dec if dec.max_len >= seq else None
        if d2 is None:
            d2 = object.__new__(Dm.LlamaDecoder)
            d2.__dict__.update(dec.__dict__)
            d2.max_len = seq
            d2._init_runtime()
        prompt = torch.randint(0, dec.s.vocab, (seq,), device=dev)
        with torch.no_grad():
            d2.prefill(prompt)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            d2.prefill(prompt)
            b.record()
            torch.cuda.synchronize()
        ttft = round(a.elapsed_time(b), 2)
        del d2
    except Exception as e:  # an extra of an extra
        ttft = repr(e)
    return {"ttft_ms_one_2048_token_prompt_whole_model": ttft,
            "workload": "Llama-2-7B E8P12, bs=%d x seq=%d prefill: the 7 QuantLinear forwards of one decoder block "
                        "(batch Hadamard kernels around %s), M=%d rows" % (
                            batch, seq, {"decompress_gemm": "decompress + dense fp16 GEMM (hipBLASLt): the default at this M, "
                                         "faster than the fused kernel here", "fused_gemm": "the fused dequant MFMA GEMM"}.get(
                                mods[0].codebook.batched_regime(M, mods[0].q_out_features, mods[0].q_in_features), "?"), M),
            "gemm_path": both["decompress_gemm"]["gemm_path"],
            "decompress_gemm": both["decompress_gemm"], "fused": both["fused"],
            "ms_per_block": round(ms, 3), "tflops": round(flops / ms / 1e9, 1),
            "roofline": {"bound": "mfma", "achieved": round(flops / ms / 1e9, 1), "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": round(flops / ms / 1e9 / 2500.0, 4)},
            "ttft_linear_layers_ms": round(ms * dec.s.layers, 1)}


def parity_bound_ulps(layers):
    """bound of decode_parity_check, fp16 ulps of rms(logits): 2 x (4 sqrt(layers) + 2).  Each of the two steps sits within
    4 sqrt(L) + 2 of the float64 model -- <= 4 ulps per block (the module bound, oracle.ulp_bound) adding in quadrature, + 2 for
    the final norm and the fp16 logits; tests/test_gpu_decode.py holds the launch to it on 1 and 8 full-size blocks, tests/
    test_gpu_block_engine_gqa.py on 1 and 4 -- so they sit within twice that of each other."""
    import math
    return 2.0 * (4.0 * math.sqrt(layers) + 2.0)


def _ulps_of_rms(delta_max, rms):
    """|delta| in fp16 ulps of rms: one ulp = 2^-10 of the power of two at or below rms (oracle.ulp_bound's unit)"""
    import math
    return float(delta_max) / (2.0 ** (math.floor(math.log2(max(float(rms), 1e-30))) - 10))


def decode_parity_check(dec, n_tokens=8):
    """Outside the timed region, TEACHER FORCED: the same `n_tokens` token ids go through (a) the captured step the bench times
    (persistent block launch / fused, grouped stages) and (b) the eager step with every fusion switched off (one launch per stage
    and module group: the reference's op sequence per QuantLinear) on the SAME model, and the logits of every position are
    compared: max |a - b| in fp16 ulps of rms(logits of b).  Bound: parity_bound_ulps(layers) = 2 (4 sqrt(L) + 2) = 49 for 32
    blocks, 76 for 80 (measured on the final kernels: profiles/README.md).  The launch differs from the unfused step in the
    order of the transforms' additions, in its block exponents (norm bound instead of the maximum) and in the fp16 rounding
    of the MLP's rows: perturbations of 1e-5 that flip fp16 roundings of module outputs in ~1 % of the elements, which the
    blocks behind carry on -- a bug (a wrong digit, row or sign) shows as hundreds of ulps.  A free-running greedy comparison is reported too but is NOT the criterion: on a random-init model two arg-maxima within an
    ulp of each other part the sequences without anything being wrong (profiles/README.md, r04f).  The line is not printed
    when the bound fails (main() raises)."""
    import torch
    with torch.no_grad():
        fused = dec.generate(n_tokens, first_token=7, use_graph=True).cpu().tolist()
        forced = [7] + fused[:-1]                       # the ids both steps are fed (position t gets forced[t])

        def run_forced(use_graph):
            dec.reset(first_token=forced[0])
            outs = []
            for t in range(n_tokens):
                dec.tok.fill_(forced[t])
                if use_graph and dec.graph is not None:
                    dec.graph.replay()
                    lg = dec.step_logits
                else:
                    lg = dec.step()
                outs.append(lg.float().reshape(-1).clone())
            return torch.stack(outs)
        la = run_forced(True)
        saved = (dec.fused_prologue, dec.chain, getattr(dec, "block_eng", False), getattr(dec, "ffn_eng", False))
        try:
            dec.fused_prologue, dec.chain = False, False      # no fused transforms / grouped GEMVs,
            dec.block_eng = dec.ffn_eng = False               # no persistent launch: one launch per stage
            lb = run_forced(False)
            plain = dec.generate(n_tokens, first_token=7, use_graph=False).cpu().tolist()
        finally:
            dec.fused_prologue, dec.chain, dec.block_eng, dec.ffn_eng = saved
    finite = bool(torch.isfinite(la).all() and torch.isfinite(lb).all())
    per_pos = []
    for t in range(n_tokens):
        rms = lb[t].pow(2).mean().sqrt().item()
        per_pos.append(round(_ulps_of_rms((la[t] - lb[t]).abs().max().item(), rms), 3))
    max_ulps = max(per_pos) if finite else float("inf")
    bound = parity_bound_ulps(dec.s.layers)
    first_diff = next((i for i, (x, y) in enumerate(zip(fused, plain)) if x != y), None)
    return {"tokens": n_tokens, "teacher_forced": True, "max_ulps": max_ulps, "max_ulps_per_position": per_pos,
            "bound_ulps": round(bound, 1), "unit": "fp16 ulps of rms(logits)", "ok": bool(finite and max_ulps <= bound),
            "argmax_agree_teacher_forced": int((la.argmax(1) == lb.argmax(1)).sum().item()),
            "captured_step": "persistent block launch" if saved[2] else ("stage-wise + MLP launch" if saved[3] else "stage-wise"),
            "engine_status": dec.engine_status() if hasattr(dec, "engine_status") else 0,
            "free_running_greedy_equal": fused == plain, "free_running_first_difference": first_diff,
            "first_tokens": fused, "first_tokens_unfused": plain}


def time_decoder(D, shape, codebook, steps, warmup, device, **cb_kwargs):
    """tokens/s of one more configuration (BASELINE configs[2], [3]) with the same procedure as the headline"""
    import torch
    big = shape.hidden >= 8192           # (70B: also timed at position 2000 of its cache below)
    # (70B: ONE copy of the codes -- the tiled one the 8192-wide launch streams)
    dec = D.LlamaDecoder(shape, codebook, max_len=(2048 if big else 0) + steps + warmup + 8, device=device, seed=0,
                         device_init=big, single_copy=big, **cb_kwargs)
    dec.capture()
    dec.reset(first_token=1)
    import gc
    gc.collect()            # (cyclic garbage of an earlier decoder freed INSIDE a timed region is a device-wide stall)
    with torch.no_grad():
        for _ in range(warmup):
            dec.graph.replay()
        torch.cuda.synchronize()
        # four quarters, their median x 4: one stall of the box (seen once: 45 ms inside 64 steps) does not become the
        # figure of an extra (the headline is timed as the contract says: K steps, one clock)
        q = max(1, steps // 4)
        dts = []
        for _ in range(4):
            t0 = time.perf_counter()
            for _ in range(q):
                dec.graph.replay()
            torch.cuda.synchronize()
            dts.append(time.perf_counter() - t0)
        steps = 4 * q
        sd = sorted(dts)
        dt = 4 * 0.5 * (sd[1] + sd[2])                  # the median of the four quarters
    algo = dec.algorithmic_bytes_per_token()
    out = {"codebook": codebook, "layers": shape.layers, "hidden": shape.hidden, "ffn": shape.ffn,
           "tokens_per_s": round(steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
           "tokens_per_s_quarters_min_max": [round(q / sd[3], 2), round(q / sd[0], 2)],
           "warmup": warmup, "algorithmic_bytes_per_token": algo,
           "token_roofline_frac": round(steps / dt / (HBM_PEAK_GBPS * 1e9 / algo), 4)}
    if codebook == "E8P12":
        out["gemv_roofline"] = engine_roofline(dec) if getattr(dec, "block_eng", False) else gemv_roofline(dec)
    tiled_only = all(L_[k].Qidxs is None for L_ in dec.layers for k in ("q", "k", "v", "o", "gate", "up", "down"))
    code_b = sum(qidxs_nbytes(L_[k]) for L_ in dec.layers for k in ("q", "k", "v", "o", "gate", "up", "down"))
    out["resident_weights"] = {"code_bytes": code_b, "copies_of_the_codes": 1 if (tiled_only or getattr(dec, "eng_shape", 0) != 1) else 2,
                               "layout": "tiled (quip_tile_codes) only: LlamaDecoder(single_copy=True)" if tiled_only else "checkpoint (row major)"}
    if big:
        # the same captured step with the position counter moved to 2000 (attention over 2000 cached rows per head)
        with torch.no_grad():
            dec.reset(first_token=1)
            dec.pos.fill_(2000)
            for _ in range(3):
                dec.graph.replay()
            dec.pos.fill_(2000)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(16):
                dec.graph.replay()
            torch.cuda.synchronize()
            dtl = time.perf_counter() - t0
        out["position_2000"] = {"tokens_per_s": round(16 / dtl, 2), "ms_per_step": round(dtl / 16 * 1e3, 4),
                                "engine_status": dec.engine_status() if hasattr(dec, "engine_status") else 0}
    if codebook == "E8P12" and shape.hidden == 8192:
        if getattr(dec, "block_eng", False) and getattr(dec, "eng_shape", 0) == 1:
            try:
                out["gemv_phases_in_launch"] = gqa_phase_rates(dec)
            except Exception as e:
                out["gemv_phases_in_launch"] = {"error": repr(e)[:300]}
            try:
                out["gemv_stream_in_launch"] = gqa_stream_rate(dec)
            except Exception as e:
                out["gemv_stream_in_launch"] = {"error": repr(e)[:300]}
        # the north star's target shapes, timed here so that the driver's run holds them (SURVEY 8d layer micro-bench)
        del dec
        torch.cuda.empty_cache()
        out["per_shape"] = gemv_per_shape([(28672, 8192), (8192, 28672), (8192, 8192), (1024, 8192), ((28672, 28672), 8192), ((8192, 1024, 1024), 8192)], device)
        return out
    del dec
    torch.cuda.empty_cache()
    return out


def hf_static_cache_extra(D, device, new_tokens=128, cache_len=2048):
    """the reference's own harness shape on the headline model (example_generate.py:62-70, 103-110): random-init HF Llama-2-7B
    -> QuipQuantizer.convert_model (every block projection a QuantLinear, E8P12) -> StaticCache(2048) -> greedy `new_tokens`
    tokens, timed around the decode loop as the reference does: (i) eager, (ii) the single-token step captured in a hipGraph
    (the reference: torch.compile(mode="reduce-overhead", fullgraph=True)); beside them LlamaDecoder.from_hf on the SAME
    module objects with a 2048-slot cache.  The reference's published 138-184 tok/s (RTX 4090) is figure (ii)'s counterpart."""
    import torch
    from transformers import AutoModelForCausalLM, LlamaConfig
    from quip_for_all_amd.quantizer import QuipQuantizer
    from quip_for_all_amd.qlinear import QuantLinear
    from quip_for_all_amd.hf_static import HFStaticDecoder
    cfg = LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                      num_key_value_heads=32, vocab_size=32000, max_position_embeddings=4096, rms_norm_eps=1e-5)
    with torch.device("meta"):
        model = AutoModelForCausalLM.from_config(cfg, dtype=torch.float16)
    qz = QuipQuantizer(codebook="E8P12", inference=True, ft_epochs=0)
    qz.convert_model(model)
    # (to_empty replaces EVERY tensor by uninitialised memory, also the real ones convert_model made -- the codebooks' tables)
    real = {n: t.detach().clone() for n, t in list(model.named_parameters()) + list(model.named_buffers()) if not t.is_meta}
    model.to_empty(device=device)
    with torch.no_grad():
        live = dict(list(model.named_parameters()) + list(model.named_buffers()))
        for n, t in real.items():
            live[n].copy_(t)
    g = torch.Generator(device=device).manual_seed(0)
    with torch.no_grad():
        for name, prm in list(model.named_parameters()) + list(model.named_buffers()):
            if prm.is_floating_point() and "inv_freq" not in name:
                prm.copy_((torch.randn(prm.shape, generator=g, device=device, dtype=torch.float32) * 0.02).to(prm.dtype))
        for m in model.modules():
            if isinstance(m, QuantLinear):
                m.Qidxs.copy_(torch.randint(-32768, 32768, m.Qidxs.shape, generator=g, device=device, dtype=torch.int32).to(m.Qidxs.dtype))
                m.SU.copy_((torch.randint(0, 2, m.SU.shape, generator=g, device=device) * 2 - 1).half())
                m.SV.copy_((torch.randint(0, 2, m.SV.shape, generator=g, device=device) * 2 - 1).half())
                m.Wscale.fill_(1.0 / 64.0)
                for nm in ("had_left", "had_right"):
                    h = getattr(m, nm)
                    if h is not None:
                        h.copy_(torch.linalg.qr(torch.randn(h.shape, generator=g, device=device))[0].half())
            if m.__class__.__name__.endswith("RMSNorm"):
                m.weight.fill_(1.0)
        # buffers computed in __init__ (the rotary frequencies) were built on the meta device: build them again
        rot = model.model.rotary_emb
        model.model.rotary_emb = type(rot)(config=cfg, device=device)
        for m in model.modules():
            if isinstance(m, QuantLinear):
                m.wscale_float = float(m.Wscale.mean().item())          # load_quantized_model's post-load step (quantizer.py:835-844)
    model.eval()
    ids = torch.randint(1, 32000, (1, 16), generator=torch.Generator().manual_seed(1)).to(device)
    out = {"model": "random-init Llama-2-7B shape, E8P12, HF LlamaForCausalLM + StaticCache(%d)" % cache_len,
           "prompt_tokens": 16, "new_tokens": new_tokens, "timing": "decode loop of new_tokens - 1 single-token steps, synchronised at both ends"}
    toks = {}
    for mode in ("eager", "graph", "compile"):           # compile: the reference's torch.compile(mode="reduce-overhead", fullgraph=True)
        dec = HFStaticDecoder(model, max_cache_len=cache_len)
        try:
            dec.generate(ids, 8, mode)                  # warm-up (and capture / compilation)
            if mode == "compile":
                dec.generate(ids, 8, mode)
        except Exception as e:
            out["hf_%s_error" % mode] = repr(e)[:300]
            continue
        ts = []
        for _ in range(3):
            t, dt = dec.generate(ids, new_tokens, mode)
            ts.append((new_tokens - 1) / dt)
        toks[mode] = t
        out["hf_%s_tokens_per_s" % mode] = round(float(np.median(ts)), 2)
        out["hf_%s_runs" % mode] = [round(x, 2) for x in ts]
        del dec
    out["hf_graph_equals_eager"] = bool(torch.equal(toks["eager"], toks["graph"]))

    def forced_compare(step_a, sync_a, ref, steps=16):
        """teacher forced on the eager stock path's tokens: `ref` (an HFStaticDecoder on the stock forward,
        prefilled) and path `step_a(token) -> logits (1, vocab)` see the same token at every step; max |difference| in fp16
        ulps of rms(logits of the stock path), and how many arg-maxima agree"""
        worst, same, gap = 0.0, 0, 0.0
        with torch.no_grad():
            for _ in range(steps):
                lb = ref._forward(ref.tok, ref.pos)[:, -1].float()
                la = step_a(ref.tok).float().reshape(1, -1)
                rms = lb.pow(2).mean().sqrt().item()
                worst = max(worst, _ulps_of_rms((la - lb).abs().max().item(), rms))
                nxt = lb.argmax(-1, keepdim=True)
                same += int(la.argmax(-1).item() == nxt.item())
                # where the arg-maxima differ: how far below the stock step's maximum is ITS logit of the token the other path chose
                gap = max(gap, _ulps_of_rms(float(lb.max() - lb[0, int(la.argmax(-1).item())]), rms))
                ref.tok.copy_(nxt)
                ref.pos += 1
                sync_a(nxt)
        return {"steps": steps, "max_ulps_of_rms_logits": round(worst, 2), "argmax_equal": same,
                "stock_logit_of_the_other_choice_below_stock_max_ulps_of_rms": round(gap, 2)}
    if "compile" in toks:
        out["hf_compile_equals_eager"] = bool(torch.equal(toks["eager"], toks["compile"]))
        # the free-running sequences of a random-init model part ways at the first near tie (profiles/r05_near_tie_rate.txt); what the
        # compiled stock step computes, against the eager stock step on the SAME tokens:
        try:
            # (the reference's compiled function returns the TOKEN -- example_generate.py:28-33 -- so the comparison is: on the same
            #  inputs, how far below the eager step's maximum is the logit of the token the compiled step chose)
            e_, c_ = HFStaticDecoder(model, max_cache_len=cache_len), HFStaticDecoder(model, max_cache_len=cache_len)
            c_.generate(ids, 4, "compile")             # compilation + the cudagraph trees' warm-up
            e_.prefill(ids)
            c_.prefill(ids)
            worst, same, steps = 0.0, 0, 16
            with torch.no_grad():
                for _ in range(steps):
                    le = e_._forward(e_.tok, e_.pos)[:, -1].float()
                    torch.compiler.cudagraph_mark_step_begin()
                    tc = int(c_._compiled(e_.tok.clone(), c_.pos.clone()).clone().item())
                    nxt = le.argmax(-1, keepdim=True)
                    same += int(tc == int(nxt.item()))
                    worst = max(worst, _ulps_of_rms(float(le.max() - le[0, tc]), le.pow(2).mean().sqrt().item()))
                    for h_ in (e_, c_):
                        h_.tok.copy_(nxt)
                        h_.pos += 1
            out["hf_compile_teacher_forced_vs_eager"] = {"steps": steps, "argmax_equal": same,
                                                         "eager_logit_of_compiled_choice_below_eager_max_ulps_of_rms": round(worst, 2)}
            del e_, c_
        except Exception as e:
            out["hf_compile_teacher_forced_vs_eager"] = {"error": repr(e)[:300]}
    # the SAME harness on the SAME model object with hf_fast.enable_fast_decode (what load_quantized_model switches on):
    # single-token calls on the StaticCache run LlamaDecoder.step() on the cache's own tensors
    try:
        from quip_for_all_amd.hf_fast import enable_fast_decode, disable_fast_decode
        enable_fast_decode(model)
        for mode in ("eager", "graph", "compile"):
            dec = HFStaticDecoder(model, max_cache_len=cache_len)
            dec.generate(ids, 8, mode)
            if mode == "compile":                      # (the step through quip_lib::hf_decode_step: fullgraph=True holds)
                dec.generate(ids, 8, mode)
            ts = []
            for _ in range(3):
                t, dt = dec.generate(ids, new_tokens, mode)
                ts.append((new_tokens - 1) / dt)
            out["hf_fast_decode_%s_tokens_per_s" % mode] = round(float(np.median(ts)), 2)
            out["hf_fast_decode_%s_runs" % mode] = [round(x, 2) for x in ts]
            # (free-running greedy sequences of a random-init model part ways at the first near tie and never meet again: this
            #  count says where that happened, the teacher-forced comparison below says how far apart the two paths are)
            out["hf_fast_decode_%s_free_running_tokens_equal_to_stock" % mode] = "%d / %d" % (int((t == toks["eager"]).sum()), new_tokens)
            del dec
        fd = model._quip_fast_decode
        # teacher-forced on the stock path's tokens: the two paths see the same inputs at every step
        a_, b_ = HFStaticDecoder(model, max_cache_len=cache_len), HFStaticDecoder(model, max_cache_len=cache_len)
        a_.prefill(ids)
        b_.prefill(ids)
        same, worst = 0, 0.0
        with torch.no_grad():
            for _ in range(32):
                model.forward = fd.orig_forward
                lb = b_._forward(b_.tok, b_.pos).float()
                model.forward = fd
                la = a_._forward(a_.tok, a_.pos).float()
                worst = max(worst, float((la - lb).abs().max() / lb.abs().max()))
                nxt = lb[:, -1].argmax(-1, keepdim=True)
                same += int(la[:, -1].argmax(-1).item() == nxt.item())
                for h_ in (a_, b_):
                    h_.tok.copy_(nxt)
                    h_.pos += 1
        out["hf_fast_decode_teacher_forced"] = {"steps": 32, "argmax_equal_to_stock": same,
                                                "max_abs_logit_difference_over_max_abs_logit": round(worst, 5)}
        del a_, b_
        out["hf_fast_decode_step"] = ("persistent block launch" if getattr(fd.dec, "block_eng", False) else "stage-wise") \
            if fd.dec is not None else "refused: %s" % fd.disabled
        # HF's own loop: model.generate(cache_implementation="static"), greedy, timed whole (prompt pass included)
        model.generation_config.eos_token_id = None
        model.generation_config.pad_token_id = 0
        model.generate(ids, max_new_tokens=8, do_sample=False, cache_implementation="static")
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.generate(ids, max_new_tokens=new_tokens, do_sample=False, cache_implementation="static")
            torch.cuda.synchronize()
            ts.append(new_tokens / (time.perf_counter() - t0))
        # (HF compiles the forward there and passes an attention mask: the wrapper leaves such traced calls to the stock forward)
        out["hf_generate_api_static_cache_tokens_per_s"] = round(float(np.median(ts)), 2)
        # ... and with generate()'s default DynamicCache (the wrapper decodes on its own static buffers and hands views back)
        model.generate(ids, max_new_tokens=8, do_sample=False)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.generate(ids, max_new_tokens=new_tokens, do_sample=False)
            torch.cuda.synchronize()
            ts.append(new_tokens / (time.perf_counter() - t0))
        out["hf_generate_api_default_cache_fast_decode_tokens_per_s"] = round(float(np.median(ts)), 2)
        disable_fast_decode(model)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.generate(ids, max_new_tokens=new_tokens, do_sample=False)
        torch.cuda.synchronize()
        out["hf_generate_api_default_cache_stock_tokens_per_s"] = round(new_tokens / (time.perf_counter() - t0), 2)
    except Exception as e:
        out["hf_fast_decode_error"] = repr(e)[:300]
    fast = D.LlamaDecoder.from_hf(model, max_len=cache_len)
    fast.generate(8, prompt=ids[0])
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ft = fast.generate(new_tokens, prompt=ids[0])
        torch.cuda.synchronize()
        ts.append(new_tokens / (time.perf_counter() - t0))
    out["llamadecoder_from_hf_tokens_per_s"] = round(float(np.median(ts)), 2)
    out["llamadecoder_from_hf_runs"] = [round(x, 2) for x in ts]
    out["llamadecoder_step"] = "persistent block launch" if getattr(fast, "block_eng", False) else "stage-wise"
    n_same = int((ft[:new_tokens].cpu() == toks["graph"].cpu()).sum())
    out["llamadecoder_free_running_tokens_equal_to_hf"] = "%d / %d" % (n_same, new_tokens)
    # ... and teacher forced: LlamaDecoder.from_hf's step against the eager stock forward on the same tokens
    try:
        e_ = HFStaticDecoder(model, max_cache_len=cache_len)
        e_.prefill(ids)
        fast.reset(first_token=int(ids[0, 0]))
        fast.prefill(ids[0])

        def f_step(tok):
            fast.tok.copy_(tok.view_as(fast.tok))
            return fast.step()
        out["llamadecoder_teacher_forced_vs_hf_eager"] = forced_compare(f_step, lambda nxt: None, e_)
        del e_
    except Exception as e:
        out["llamadecoder_teacher_forced_vs_hf_eager"] = {"error": repr(e)[:300]}
    return out


def long_context_decode(D, device, positions=(2048, 4000), steps=16):
    """the headline model's decode step at long contexts (same captured step, position counter moved: the cache rows hold
    whatever earlier tokens left there -- attention time does not depend on the values)"""
    import torch
    dec = D.LlamaDecoder(D.LLAMA2_7B, "E8P12", max_len=max(positions) + steps + 8, device=device, seed=0, device_init=True)
    dec.capture()
    out = {}
    with torch.no_grad():
        for p in positions:
            dec.reset(first_token=1)
            dec.pos.fill_(p)
            for _ in range(3):
                dec.graph.replay()
            dec.pos.fill_(p)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                dec.graph.replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out["position_%d" % p] = {"tokens_per_s": round(steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4)}
    out["engine_status"] = dec.engine_status() if hasattr(dec, "engine_status") else 0
    out["step"] = "persistent block launch" if getattr(dec, "block_eng", False) else "stage-wise"
    del dec
    torch.cuda.empty_cache()
    return out


def max_over_ranks(dist, dt, device):
    """the job's step time is the slowest replica's"""
    if dist is None:
        return dt
    tt = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def aggregate_tokens_per_s(world, steps, dt):
    """whole-job throughput of `world` independent replicas that each decoded `steps` tokens"""
    return world * steps / dt


def dist_selftest(a, rank, world):
    """replica plumbing only, no GPU: every rank "decodes" for a rank-dependent synthetic time"""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    dt = 0.001 * a.steps * (rank + 1)          # rank r needs (r + 1) ms per token
    dt = max_over_ranks(dist if world > 1 else None, dt, "cpu")
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "steps": a.steps,
                          "value": round(aggregate_tokens_per_s(world, a.steps, dt), 4),
                          "ms_per_step": round(dt / a.steps * 1e3, 4), "scaling": "weak"}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--model", default="7b", choices=["7b", "70b", "tiny"])
    ap.add_argument("--codebook", default="E8P12")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prefill", action="store_true", help="skip the configs[4] prefill extra of the N=1 7B line")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the configs[2] / [3] extras (70B E8P12, 7B E8P12RVQ4B, 7B D4) of the N=1 7B line")
    ap.add_argument("--dist-selftest", action="store_true",
                    help="exercise only the replica plumbing (rendezvous, barrier, max-over-ranks, rank-0 line) "
                         "with a synthetic per-rank time; runs on CPU with gloo (tests/test_bench_replicas.py)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.dist_selftest:
        return dist_selftest(a, rank, world)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback of the hot path)"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        dist = dist_mod

    if not os.path.exists(os.path.join(REPO, "quip_for_all_amd", "lib", "libquip_mi355.so")):
        import __graft_entry__ as G      # build product missing from this checkout: compile it (rank 0 first)
        if rank == 0:
            G.build()
        if dist is not None:
            dist.barrier()
    import quip_for_all_amd  # noqa: F401
    from quip_for_all_amd import decode as D
    shape = {"7b": D.LLAMA2_7B, "70b": D.LLAMA2_70B, "tiny": D.TINY}[a.model]
    # a 2048-slot static cache, as the reference's harness sets up (example_generate.py:66); the timed tokens sit at
    # positions [warmup, warmup + steps) of it in every repeat (config.workload says so)
    max_len = max(2048, a.steps + a.warmup + 8)
    dec = D.LlamaDecoder(shape, a.codebook, max_len=max_len, device=f"cuda:{local_rank}", seed=rank,
                         device_init=(a.model == "70b"))
    dec.capture()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    import gc
    gc.collect()
    gc.disable()            # no collector pause (and no freeing of device memory) inside the timed region
    # the contract's region -- W untimed steps, then EXACTLY K steps between two barriers -- three times over (same start
    # state each time); `value` is the median region, `runs_tokens_per_s` all three
    dts = []
    for rep in range(3):
        dec.reset(first_token=1 + rank)
        with torch.no_grad():
            for _ in range(a.warmup):
                dec.graph.replay()
        barrier()
        t0 = time.perf_counter()
        with torch.no_grad():
            for _ in range(a.steps):
                dec.graph.replay()
        barrier()
        dts.append(max_over_ranks(dist, time.perf_counter() - t0, f"cuda:{local_rank}"))
    gc.enable()
    dt = sorted(dts)[1]
    status = dec.engine_status() if hasattr(dec, "engine_status") else 0
    if status:      # a persistent launch gave up on a hand-off: its tokens are not results, no line
        raise RuntimeError("rank %d: persistent decode launch gave up (code 0x%x)" % (rank, status))

    if rank == 0:
        tok_s = aggregate_tokens_per_s(world, a.steps, dt)
        algo_bytes = dec.algorithmic_bytes_per_token()
        out = {
            "metric": METRIC, "value": round(tok_s, 2), "unit": "tokens/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "i8xi8->i32 (fp16 I/O)", "data": "synthetic",
            "runs_tokens_per_s": [round(aggregate_tokens_per_s(world, a.steps, d), 2) for d in dts],
            "config": {"workload": "Llama-2-%s %s random-init, bs=1 greedy decode, 1 hipGraph replay per token, positions "
                                   "[%d, %d) of a %d-slot static KV cache" % (a.model.upper(), a.codebook, a.warmup,
                                                                              a.warmup + a.steps, max_len),
                       "context": {"cache_slots": max_len, "first_timed_position": a.warmup, "last_timed_position": a.warmup + a.steps - 1},
                       "layers": shape.layers, "hidden": shape.hidden, "ffn": shape.ffn, "vocab": shape.vocab,
                       "parallelism": "replicas x%d (no collective on the data path)" % world},
            "token_roofline": {"algorithmic_bytes_per_token": algo_bytes,
                               "tokens_per_s_at_8TBps": round(HBM_PEAK_GBPS * 1e9 / algo_bytes, 1),
                               "frac": round(tok_s / world / (HBM_PEAK_GBPS * 1e9 / algo_bytes), 4)},
        }
        if a.codebook == "E8P12":
            out["roofline"] = engine_roofline(dec) if getattr(dec, "block_eng", False) else gemv_roofline(dec)
        try:
            out["parity"] = decode_parity_check(dec)
        except Exception as e:      # (a transient failure of the check -- e.g. the launch giving a hand-off up on a shared
            out["parity"] = {"ok": False, "error": repr(e)[:400]}          # device -- is reported as a failed check, not as a traceback)
        if not out["parity"]["ok"]:      # a fast step whose logits differ from the unfused step's is not a result: no line
            raise RuntimeError("parity check failed: %r" % (out["parity"],))
        if a.model == "7b" and a.codebook == "E8P12" and world == 1 and not a.no_prefill:
            try:
                out["prefill"] = prefill_config5(dec)
            except Exception as e:  # an extra, never the headline
                out["prefill"] = {"error": repr(e)}
        if a.model == "7b" and a.codebook == "E8P12" and world == 1 and not a.no_extras:
            # BASELINE configs[2] and [3], timed by the same procedure after the headline (never part of `value`)
            del dec
            torch.cuda.empty_cache()
            extras = {}
            for key, (shp, cbk, st, kw) in {"llama2_70b_e8p12": (D.LLAMA2_70B, "E8P12", 32, {}),
                                            "llama2_7b_e8p12rvq4b": (D.LLAMA2_7B, "E8P12RVQ4B", 64, {}),
                                            "llama2_7b_e8p12rvq3b": (D.LLAMA2_7B, "E8P12RVQ3B", 64, {}),
                                            "llama2_7b_d4": (D.LLAMA2_7B, "D4", 64, {}),
                                            "llama2_7b_hi": (D.LLAMA2_7B, "HI", 64, {}),
                                            # the grouped-query 4096-wide shape (round 5: the shape-2 persistent launch, decode_block_g8.hip)
                                            "llama3_8b_shape_e8p12": (D.LLAMA3_8B, "E8P12", 64, {})}.items():
                try:
                    extras[key] = time_decoder(D, shp, cbk, st, 8, f"cuda:{local_rank}", **kw)
                except Exception as e:
                    extras[key] = {"error": repr(e)}
            try:
                extras["llama2_7b_e8p12_long_context"] = long_context_decode(D, f"cuda:{local_rank}")
            except Exception as e:
                extras["llama2_7b_e8p12_long_context"] = {"error": repr(e)}
            try:
                extras["hf_generate_static_cache"] = hf_static_cache_extra(D, f"cuda:{local_rank}")
            except Exception as e:
                extras["hf_generate_static_cache"] = {"error": repr(e)[:400]}
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out["extras"] = extras
        if not a.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only (bench contract)
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the checker must never take the bench down
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
