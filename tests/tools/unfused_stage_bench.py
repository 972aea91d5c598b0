""""achieved GB/s for composite / any unfused stage".  The standalone boundary kernels of the unfused seam are
HBM-bound; this prints, for each, the ALGORITHMIC bytes of one call (what the call must read and write once), the HIP-event
mean duration of the launch and the GB/s that follow, at sizes large enough to leave the caches (tens of thousands of rays).
The product path never launches them (it composites inside the fused kernel); they serve callers of raw2outputs() / Embedder.

    python tests/tools/unfused_stage_bench.py            (MI355X; summary kept in profiles/rNN_unfused_stages.txt)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import cfnerf_amd                                    # noqa: F401  (package alias for the hyphenated directory)
from cfnerf_amd import _lib as L

lib = L.lib()
HBM_PEAK_GBS = 8000.0                                 # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def ev_ms(fn, n=30, warm=40):                        # (the first launches after an allocation run at ramping clocks)
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def line(name, shape, nbytes, ms):
    gbs = nbytes / (ms * 1e-3) / 1e9
    print(f"{name:34s} {shape:28s} {nbytes / 1e6:9.1f} MB  {ms * 1e3:9.1f} us  {gbs:7.0f} GB/s  {gbs / HBM_PEAK_GBS:5.2f} of 8 TB/s", flush=True)


def main():
    dev = "cuda"
    S = 128
    # what this box's memory system gives a trivial kernel (context for the write-heavy stages below): torch's fill and copy over 1 GiB
    a, b = torch.empty(1 << 28, device=dev), torch.empty(1 << 28, device=dev)
    line("(context) torch fill_, 1 GiB written", "", a.numel() * 4, ev_ms(lambda: a.fill_(1.0)))
    line("(context) torch copy_, 1 + 1 GiB", "", 2 * a.numel() * 4, ev_ms(lambda: b.copy_(a)))
    line("(context) torch sum, 1 GiB read", "", a.numel() * 4, ev_ms(lambda: a.sum()))
    del a, b
    print(f"{'kernel':34s} {'shape':28s} {'algorithmic':>12s}  {'launch':>12s}  {'achieved':>12s}")
    for (N, K) in ((65536, 4), (16384, 32), (8192, 64)):
        raw = torch.randn(N, S, K, 4, device=dev)
        z = torch.sort(torch.rand(N, S, device=dev), -1).values.contiguous()
        d = torch.randn(N, 3, device=dev)
        rgb, disp, depth = torch.empty(N, 3, K, device=dev), torch.empty(N, K, device=dev), torch.empty(N, K, device=dev)
        wts = torch.empty(N, S, K, device=dev)
        shape = f"N={N} S={S} K={K}"
        # raw2outputs without / with the weights output (RUN:411-454): 16 S K + 4 S + 12 in, 20 K (+ 4 S K) out per ray
        b_in, b_out = N * (16 * S * K + 4 * S + 12), N * 20 * K
        ms = ev_ms(lambda: L.check(lib.cfnerf_composite_fwd(L.ptr(raw), L.ptr(z), L.ptr(d), N, S, K, 0, L.ptr(rgb), L.ptr(disp), L.ptr(depth),
                                                           None, L.stream()), "composite_fwd"))
        line("composite_kernel", shape, b_in + b_out, ms)
        ms = ev_ms(lambda: L.check(lib.cfnerf_composite_fwd(L.ptr(raw), L.ptr(z), L.ptr(d), N, S, K, 0, L.ptr(rgb), L.ptr(disp), L.ptr(depth),
                                                           L.ptr(wts), L.stream()), "composite_fwd"))
        line("composite_kernel (+weights)", shape, b_in + b_out + N * 4 * S * K, ms)
        # its adjoint, stateless: reads raw, z, d and the cotangents of the three maps, writes d_raw
        d_rgb, d_disp, d_depth = torch.randn(N, 3, K, device=dev), torch.randn(N, K, device=dev), torch.randn(N, K, device=dev)
        d_raw = torch.empty_like(raw)
        ms = ev_ms(lambda: L.check(lib.cfnerf_composite_bwd(L.ptr(raw), L.ptr(z), L.ptr(d), N, S, K, 0, L.ptr(d_rgb), L.ptr(d_disp), L.ptr(d_depth),
                                                           None, L.ptr(d_raw), L.stream()), "composite_bwd"))
        line("composite_bwd_kernel", shape, b_in + N * 20 * K + N * 16 * S * K, ms)
        del raw, wts, d_raw
    # Embedder (HLP:21-69): 12 bytes in, 4 * (3 + 6 multires) out per point
    for (P, mr) in ((1 << 23, 10), (1 << 23, 4)):
        x = torch.randn(P, 3, device=dev)
        out = torch.empty(P, 3 + 6 * mr, device=dev)
        ms = ev_ms(lambda: L.check(lib.cfnerf_embed(L.ptr(x), P, mr, L.ptr(out), L.stream()), "embed"))
        line("embed_kernel", f"P={P} multires={mr}", P * (12 + 4 * (3 + 6 * mr)), ms)
    # stratified sampling + points (RUN:510-534): 44 B per ray and 4 B of t_rand per sample in, 4 (z) + 12 (pts) B per sample out
    N, S = 1 << 18, 128
    rays = torch.randn(N, 11, device=dev); rays[:, 6] = 0.0; rays[:, 7] = 1.0
    t_vals = torch.linspace(0, 1, S, device=dev); t_rand = torch.rand(N, S, device=dev)
    z, pts = torch.empty(N, S, device=dev), torch.empty(N, S, 3, device=dev)
    ms = ev_ms(lambda: L.check(lib.cfnerf_sample_points(L.ptr(rays), L.ptr(t_vals), L.ptr(t_rand), 0, N, S, L.ptr(z), L.ptr(pts), L.stream()), "sample"))
    line("sample_points_kernel", f"N={N} S={S}", N * (44 + S * 20), ms)


if __name__ == "__main__":
    main()
