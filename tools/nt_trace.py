"""Phase timeline of the tiled NT GEMM launches INSIDE the LTX step (round 6; the review's "find out where the lab-to-step gap goes with stamps, not estimates").

Needs the trace build of the library (never the product one):
    FTMI_TRACE=1 python -m finetrainers_amd.csrc.build          -> finetrainers_amd/libftmi355_trace.so
    FTMI_LIB_PATH=finetrainers_amd/libftmi355_trace.so python tools/nt_trace.py [out_prefix] [--lab]

Thread 0 of every workgroup stamps s_memrealtime (100 MHz, one clock for the whole chip) and s_memtime (shader cycles) at its phase boundaries
(gemm.hip NT_STAMP: 0 entry, 1 K loop called, 2 first stage landed, 3 / 4 LoRA mid-round, 5 K loop done, 6 last store issued).  Per launch class
(tile, epilogue, extension, shape) this prints, averaged over the class's launches of ONE steady-state step:
    span       first workgroup entry -> last workgroup's last store issue (what rocprofv3 calls the kernel's duration, minus the drain)
    skew       last entry - first entry of the first round of workgroups (launch ramp)
    cold       entry -> first stage landed + fragments read, mean over the workgroups of the first round (the exposed cold start)
    kloop      first stage landed -> accumulators final, mean per workgroup;  cyc/stage = the same in shader cycles per K = 64 stage
    mid        the LoRA mid-round (bf16 re-rounding of the base accumulators) where there is a K-extension
    epi        accumulators final -> last store issued, mean per workgroup
    clk        shader clock during the K loop (cycles / wall time)
`--lab` runs the same launch classes stand-alone afterwards (same shapes through ftmi_gemm_nt on rotating weight copies, warm activations) so that the two
columns can be compared phase by phase on one box."""
import collections
import math
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def read_trace(prefix):
    recs = []
    with open(prefix + ".meta") as f:
        for line in f:
            v = [int(x) for x in line.split()]
            recs.append(dict(zip(("bm", "bn", "mfma16", "epi", "ext", "M", "N", "K", "K2", "nwg", "off"), v)))
    raw = open(prefix + ".bin", "rb").read()
    data = struct.unpack(f"<{len(raw) // 8}Q", raw)
    return recs, data


def summarise(recs, data, title, out):
    groups = collections.OrderedDict()
    for r in recs:
        key = (r["bm"], r["bn"], "16x16x32" if r["mfma16"] else "32x32x16", ("store", "gelu", "resid", "dgelu")[r["epi"]], r["ext"], r["M"], r["N"], r["K"], r["K2"])
        wgs = []
        for w in range(r["nwg"]):
            b = r["off"] + w * 16
            st = data[b:b + 16]
            if st[0] == 0:  # a padded workgroup that left at once
                continue
            wgs.append(st)
        if not wgs:
            continue
        t0 = min(s[0] for s in wgs)
        t_end = max(s[12] for s in wgs)
        first_round = sorted(wgs, key=lambda s: s[0])[: min(len(wgs), 256)]
        skew = (max(s[0] for s in first_round) - t0) * 0.01
        has2 = all(s[4] for s in wgs)
        cold = sum((s[4] - s[0]) for s in first_round) / len(first_round) * 0.01 if has2 else float("nan")
        start_k = 4 if has2 else 0
        kl = sum((s[10] - s[start_k]) for s in wgs) / len(wgs) * 0.01
        klc = sum((s[11] - s[start_k + 1]) for s in wgs) / len(wgs)
        mid = sum((s[8] - s[6]) for s in wgs) / len(wgs) * 0.01 if all(s[6] and s[8] for s in wgs) else 0.0
        midc = sum((s[9] - s[7]) for s in wgs) / len(wgs) if mid else 0.0
        epi = sum((s[12] - s[10]) for s in wgs) / len(wgs) * 0.01
        epic = sum((s[13] - s[11]) for s in wgs) / len(wgs)
        stages = (r["K"] + r["K2"]) // 64
        g = groups.setdefault(key, [])
        g.append(dict(span=(t_end - t0) * 0.01, skew=skew, cold=cold, kloop=kl, kloop_cyc=klc, mid=mid, mid_cyc=midc, epi=epi, epi_cyc=epic, n=len(wgs), stages=stages,
                      xcds=len({s[14] & 15 for s in wgs})))
    print(f"# {title}", file=out)
    print(f"# {'tile':9s} {'mfma':8s} {'epi':5s} ext  {'M x N x K(+K2)':24s} launches wgs |  span   skew   cold  kloop (cyc/stage)   mid    epi (cyc)   clk GHz", file=out)
    for key, g in groups.items():
        bm, bn, mf, epi, ext, M, N, K, K2 = key
        m = lambda k: sum(x[k] for x in g) / len(g)
        clk = m("kloop_cyc") / (m("kloop") * 1e3) if m("kloop") else float("nan")
        print(f"  {bm:3d}x{bn:3d}  {mf:8s} {epi:5s} {ext:d}    {M:5d}x{N:4d}x{K:4d}+{K2:<4d}      {len(g):4d}   {int(m('n')):4d} | {m('span'):6.1f} {m('skew'):6.2f} {m('cold'):6.2f} {m('kloop'):6.1f} ({m('kloop_cyc') / g[0]['stages']:6.0f})"
              f"  {m('mid'):5.2f}  {m('epi'):5.2f} ({m('epi_cyc'):6.0f})   {clk:5.2f}", file=out)
    out.flush()


def main():
    prefix = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "gpurun_out/nt_trace"
    lab = "--lab" in sys.argv
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    import bench
    from finetrainers_amd import _lib, ops
    from finetrainers_amd.parallel import DataParallelBackend

    lib = _lib.load()
    if not hasattr(lib, "ftmi_trace_enable"):
        raise SystemExit("this library has no trace entry points: build with FTMI_TRACE=1 and set FTMI_LIB_PATH")
    import ctypes

    lib.ftmi_trace_enable.argtypes = [ctypes.c_long]
    lib.ftmi_trace_dump.argtypes = [ctypes.c_char_p]
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    par = DataParallelBackend()
    ctx = bench._build_ltx(args, par, par.device)
    for _ in range(4):
        ctx["one_step"]()
    torch.cuda.synchronize()
    assert lib.ftmi_trace_enable(96 << 20) == 0
    ctx["one_step"]()
    torch.cuda.synchronize()
    lib.ftmi_trace_enable(0)
    n = lib.ftmi_trace_dump((prefix + "_step").encode())
    recs, data = read_trace(prefix + "_step")
    with open(prefix + ".txt", "w") as out:
        summarise(recs, data, f"in the step ({n} tiled NT launches of one steady-state optimisation step, bench.py's configuration)", out)
        if lab:
            # the same launch classes stand-alone: rotating weight copies (cold W), one activation buffer (warm X) -- what tools/bench_gemm_ab.py times
            seen = collections.OrderedDict()
            for r in recs:
                seen.setdefault((r["M"], r["N"], r["K"], r["K2"], r["epi"]), r)
            dev = par.device
            g = torch.Generator(device=dev).manual_seed(0)
            assert lib.ftmi_trace_enable(96 << 20) == 0
            for (M, N, K, K2, epi) in seen:
                if M < 1024:
                    continue
                x = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16)
                ncopy = max(2, int(6e8 // (N * K * 2)))
                ws = [(torch.randn((N, K), device=dev, generator=g) / math.sqrt(K)).to(torch.bfloat16) for _ in range(ncopy)]
                b = torch.randn((N,), device=dev, generator=g).to(torch.bfloat16)
                resid = torch.randn((M, N), device=dev, generator=g).to(torch.bfloat16)
                A = torch.randn(64, K, device=dev, generator=g) / math.sqrt(K)
                Bm = torch.randn(N, 64, device=dev, generator=g) * 0.05
                for i in range(12):
                    w = ws[i % ncopy]
                    if K2:
                        ops.linear_lora_fwd(x, w, b, A, Bm, 0.5, variant=8)  # (plain-store epilogue: the extension's cost is what this row shows)
                    elif epi == 1:
                        ops.gemm_nt(x, w, b, epilogue=_lib.EPI_GELU, want_out2=True, variant=8)
                    elif epi == 2:
                        ops.gemm_nt(x, w, b, epilogue=_lib.EPI_RESID, resid=resid, variant=8)
                    elif epi == 3:
                        ops.gemm_nt(x, w, b, epilogue=_lib.EPI_DGELU, aux=resid, variant=8)
                    else:
                        ops.gemm_nt(x, w, b, variant=8)
                torch.cuda.synchronize()
            lib.ftmi_trace_enable(0)
            lib.ftmi_trace_dump((prefix + "_lab").encode())
            recs2, data2 = read_trace(prefix + "_lab")
            summarise(recs2, data2, "stand-alone (12 back-to-back launches per class: weights rotate through > 600 MB, ONE activation buffer -- the lab's conditions)", out)
    print(open(prefix + ".txt").read())
    par.destroy()


if __name__ == "__main__":
    main()
