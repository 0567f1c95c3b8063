"""GEMM variant micro-benchmark on the GPU box: python tools/gemm_bench.py [M]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reprover_amd import _lib
if os.environ.get("LIB"):  # another build of the library (probe builds: tools/probes/gemm_phase.py build)
    _lib.LIB_PATH = os.path.abspath(os.environ["LIB"])
lib = _lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
variants = [int(v) for v in os.environ.get("VARIANTS", "26,20,9,0").split(",")]
groups = [int(v) for v in os.environ.get("GROUP_M", "8").split(",")]
staggers = [int(v) for v in os.environ.get("STAGGER", "0").split(",")]
persists = [int(v) for v in os.environ.get("PERSIST", "0").split(",")]  # gemm_persist masks (29 = every projection)  # gemm_stagger_us_* values (first-round spread)
OPT = {"wi": b"gemm_stagger_us_wi", "wo": b"gemm_stagger_us_wo", "qk": b"gemm_stagger_us_qkv", "o ": b"gemm_stagger_us_o"}
if os.environ.get("SKINNY") is not None:  # 0: the variant asked for is the variant run, whatever the token count
    _lib.check(lib.rp_set_option(b"gemm_skinny", int(os.environ["SKINNY"])), "opt")
dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(0)
only = os.environ.get('ONLY')
shapes = [("wi  geglu", 7168, 1472, _lib.RP_EPI_GEGLU_BF16), ("wo  resid", 1472, 3584, _lib.RP_EPI_RESID),
          ("qkv store", 1152, 1472, _lib.RP_EPI_STORE_BF16), ("o   resid", 1472, 384, _lib.RP_EPI_RESID)]
ROUNDS = int(os.environ.get("ROUNDS", "3"))
for name, N, K, epi in shapes:
    if only and not name.startswith(only):
        continue
    A = (torch.randn(M, K, generator=g, device=dev)).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    fill = os.environ.get("FILL", "random")  # DVFS check: the same kernel on zero / sign-constant operands
    if fill == "zero":
        A.zero_(); W.zero_()
    elif fill == "abs":
        A = A.abs(); W = W.abs()
    CH = M if M <= 4096 else 256
    ref = A[:CH].float() @ W.float().T
    if epi == _lib.RP_EPI_RESID:
        out = torch.zeros(2, M, N, dtype=torch.bfloat16, device=dev)  # the residual stream's hi / lo planes
    elif epi == _lib.RP_EPI_GEGLU_BF16:
        out = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
    else:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)

    # FUSED=1: the encoder's real epilogues (residual GEMMs also write the bf16 copy and the
    # sum-of-squares partials); default: the bare GEMM + epilogue arithmetic
    fused = os.environ.get("FUSED") == "1" and epi == _lib.RP_EPI_RESID
    np_ = (N + 63) // 64
    ssp = torch.empty(np_, M, device=dev) if fused else None

    # COLD=n: n copies of the weight matrix, used in rotation (n x N x K x 2 bytes past the 256 MB Infinity Cache: every launch
    # streams its weights from HBM, as a single-state retrieve() does - 434 MB of weights per pass)
    ncold = int(os.environ.get("COLD", "0"))
    Ws = [W] + [W.clone() for _ in range(max(0, ncold - 1))]
    turn = [0]

    def run():
        Wc = Ws[turn[0] % len(Ws)]
        turn[0] += 1
        if fused:
            _lib.check(lib.rp_dbg_gemm_fused(A.data_ptr(), Wc.data_ptr(), out.data_ptr(), M, N, K, N, epi, None, 0, 0.0,
                                             0.0, None, ssp.data_ptr(), np_, _lib.current_stream()), "gemm")
        else:
            _lib.check(lib.rp_dbg_gemm(A.data_ptr(), Wc.data_ptr(), out.data_ptr(), M, N, K, N, epi,
                                       _lib.current_stream()), "gemm")
    times = {}
    errs = {}
    for rnd in range(ROUNDS + 1):  # round 0 = correctness + warm-up; then interleaved timing rounds
        for v in variants:
            for gm in [(g_, s_, p_) for g_ in groups for s_ in staggers for p_ in persists]:
                _lib.check(lib.rp_set_option(b"gemm_variant_all", v), "opt")
                _lib.check(lib.rp_set_option(b"gemm_persist", gm[2]), "opt")
                _lib.check(lib.rp_set_option(b"gemm_group_m", gm[0]), "opt")
                if gm[1]:  # (the stagger knob exists in RP_EXPERIMENTS builds only)
                    _lib.check(lib.rp_set_option(OPT[name[:2]], gm[1]), "opt")
                if rnd == 0:
                    if epi == _lib.RP_EPI_RESID:
                        out.zero_()
                    run(); torch.cuda.synchronize()
                    if epi == _lib.RP_EPI_RESID:
                        err = (out[0, :CH].float() + out[1, :CH].float() - ref).abs().max().item()
                    elif epi == _lib.RP_EPI_STORE_BF16:
                        err = (out[:CH].float() - ref).abs().max().item()
                    else:
                        r = ref.view(CH, N // 64, 2, 32)
                        gg, uu = r[:, :, 0].reshape(CH, -1), r[:, :, 1].reshape(CH, -1)
                        want = 0.5 * gg * (1 + torch.tanh(0.7978845608 * (gg + 0.044715 * gg ** 3))) * uu
                        err = (out[:CH].float() - want).abs().max().item()
                    errs[(v, gm)] = err
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                iters = 5
                e0.record()
                for _ in range(iters):
                    run()
                e1.record(); torch.cuda.synchronize()
                times.setdefault((v, gm), []).append(e0.elapsed_time(e1) / iters)
    for (v, gm), ts in times.items():
        best, med = min(ts), sorted(ts)[len(ts) // 2]
        print(f"{name} M={M} N={N} K={K} variant={v:2d} group_m,stagger_us,persist={gm}: best {best:7.3f} ms {2.0*M*N*K/best/1e9:7.1f} TF | "
              f"median {med:7.3f} ms {2.0*M*N*K/med/1e9:7.1f} TF  maxerr {errs[(v, gm)]:.3e}", flush=True)
