"""Parity of single HIP kernels (through the C ABI op entry points) against CPU fp32 math.
GEMM / LayerNorm / attention tolerances are fp32 summation-order noise; the sampler's pre-draw
distribution is compared with the oracle's restatement of api_cache.py:169-177."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from mgea import ops as _ops
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return _ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


@pytest.mark.parametrize("M,N,K,split", [
    (64, 1536, 512, 0), (64, 512, 2048, 0), (64, 8324, 512, 0), (1, 512, 512, 0), (7, 311, 128, 4),
    (64, 512, 512, 1), (200, 384, 128, 1), (513, 1000, 96 * 4, 1), (1024, 2048, 512, 1), (130, 8324, 512, 1)])
def test_gemm_f32(ops, M, N, K, split):
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    got = ops.gemm(a.cuda(), w.cuda(), b.cuda(), split_k=split).cpu()
    want = (a.double() @ w.double().t() + b.double()).float()
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=2e-5, rtol=0)


def test_gemm_is_deterministic(ops):
    a, w = rnd(64, 2048, seed=5).cuda(), rnd(512, 2048, seed=6).cuda()
    r0 = ops.gemm(a, w)
    for _ in range(3):
        assert torch.equal(r0, ops.gemm(a, w))


def test_gemm_rejects_bad_k(ops):
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros(4, 48).cuda(), torch.zeros(8, 48).cuda())


@pytest.mark.parametrize("M,C,eps", [(64, 512, 1e-5), (5, 128, 1e-5), (300, 768, 1e-12), (3, 2048, 1e-5)])
def test_layernorm(ops, M, C, eps):
    x, w, b = rnd(M, C, seed=1, scale=3.0), rnd(C, seed=2) + 1.0, rnd(C, seed=3)
    got = ops.layernorm(x.cuda(), w.cuda(), b.cuda(), eps).cpu()
    want = torch.nn.functional.layer_norm(x, (C,), w, b, eps)
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=2e-5, rtol=0)


def ref_attention(qkv, H, valid):
    B, T, C3 = qkv.shape
    C = C3 // 3
    dh = C // H
    q, k, v = (qkv[..., i * C:(i + 1) * C].reshape(B, T, H, dh).transpose(1, 2).double() for i in range(3))
    s = q @ k.transpose(-1, -2) / dh ** 0.5
    s = s.masked_fill(~valid[:, None, None, :], float("-inf"))
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, T, C).float()


@pytest.mark.parametrize("B,T,H,dh", [(2, 5, 2, 64), (3, 64, 8, 64), (2, 130, 12, 64), (1, 200, 4, 32), (4, 24, 2, 64)])
@pytest.mark.parametrize("masking", ["none", "lens", "mask"])
def test_attention_dense(ops, B, T, H, dh, masking):
    C = H * dh
    qkv = rnd(B, T, 3 * C, seed=7, scale=1.5)
    valid = torch.ones(B, T, dtype=torch.bool)
    lens = mask = None
    if masking == "lens":
        lens = torch.tensor([max(1, T - 3 * b - 1) for b in range(B)])
        valid = torch.arange(T)[None, :] < lens[:, None]
    elif masking == "mask":
        g = torch.Generator().manual_seed(3)
        valid = torch.rand(B, T, generator=g) > 0.3
        valid[:, 0] = True
        mask = valid.to(torch.int32)
    got = ops.attention(qkv.cuda(), H, None if lens is None else lens.cuda(), None if mask is None else mask.cuda()).cpu()
    want = ref_attention(qkv, H, valid)
    rows = torch.ones(B, T, dtype=torch.bool) if lens is None else valid   # padded query rows are don't-care
    np.testing.assert_allclose(got[rows].numpy(), want[rows].numpy(), atol=2e-5, rtol=0)


@pytest.mark.parametrize("B,T,H,dh", [(2, 1024, 8, 64), (1, 1000, 8, 64)])
@pytest.mark.parametrize("masking", ["none", "lens"])
def test_attention_dense_long_sequences(ops, B, T, H, dh, masking):
    """The flash attention of the f32 prefill at the sequence lengths the [64, 1024] prefill benchmark runs it at (VERDICT r3 #1b: the
    op test used to stop at T = 200): 16 key tiles per query block, masked (ragged `lens`: 1 key, a length that is no multiple of the
    64-key tile, the whole sequence) and unmasked, every real query row against fp64 math (api_cache.py:68: MultiheadAttention, no mask)."""
    C = H * dh
    qkv = rnd(B, T, 3 * C, seed=17, scale=1.5)
    valid = torch.ones(B, T, dtype=torch.bool)
    lens = None
    if masking == "lens":
        lens = torch.tensor([[T, 517][b] if B == 2 else 1 for b in range(B)])
        valid = torch.arange(T)[None, :] < lens[:, None]
    got = ops.attention(qkv.cuda(), H, None if lens is None else lens.cuda(), None).cpu()
    want = ref_attention(qkv, H, valid)
    rows = valid
    err = float((got[rows] - want[rows]).abs().max())
    assert err < 2e-5, err


def test_sampler_probabilities_match_reference_restatement(ops):
    from oracle.decoder_ref import DecoderRef
    logits = rnd(6, 8324, seed=11, scale=3.0)
    for temp, k, p in [(1.0, 50, None), (0.8, 50, None), (1.3, 1, None), (1.0, None, 0.9), (0.7, 200, 0.5),
                       (1.0, 8324, None), (1.0, 5, 0.999)]:
        ids, probs = ops.sample(logits.cuda(), temp, k, p, seed=3, step=0, want_probs=True)
        want = DecoderRef.masked_probs(logits, temp, k, p)
        np.testing.assert_allclose(probs.cpu().numpy(), want.numpy(), atol=2e-6, rtol=1e-4)
        chosen = want.gather(1, ids.cpu().long()[:, None])
        assert bool((chosen > 0).all()), "sampled a token outside the kept set"
        if k == 1:
            assert ids.cpu().tolist() == logits.argmax(1).tolist()


def test_topk_keeps_exactly_k_on_ties(ops):
    """api_cache.py:172-175: topk + scatter_ leaves exactly top_k entries unmasked, also when several logits equal the
    k-th largest.  Which of the tied entries torch keeps is unspecified; here it is the lowest ids."""
    V, k = 700, 50
    flat = torch.zeros(1, V)                                            # everything tied
    lg = torch.full((1, V), -1.0)
    lg[0, torch.arange(5, 700, 70)] = 3.0                               # 10 clear winners ...
    lg[0, 300:420] = 1.5                                                # ... then 120 entries tied for places 11..130
    rnd = torch.randn(1, V)
    rnd[0, 100] = rnd[0, 613] = rnd[0, 7] = float(rnd.topk(k).values[0, -1])   # three-way tie exactly at the boundary
    for logits in (flat, lg, rnd):
        _, probs = ops.sample(logits.cuda(), 1.0, k, None, seed=1, step=0, want_probs=True)
        p = probs.cpu()[0]
        kept = (p > 0).nonzero().flatten().tolist()
        assert len(kept) == k, f"{len(kept)} entries survive a top-{k} cut"
        assert abs(float(p.sum()) - 1.0) < 1e-5
        kth = float(logits[0].topk(k).values[-1])
        above = (logits[0] > kth).nonzero().flatten().tolist()
        tied = (logits[0] == kth).nonzero().flatten().tolist()
        assert kept == sorted(above + tied[: k - len(above)])           # all larger ones, then ties by ascending id
        want = torch.softmax(logits[0][kept].double(), 0)
        np.testing.assert_allclose(p[kept].double().numpy(), want.numpy(), atol=1e-6)
    # top-p after such a cut still works on the k survivors
    _, probs = ops.sample(lg.cuda(), 1.0, k, 0.5, seed=1, step=0, want_probs=True)
    assert 1 <= int((probs > 0).sum()) <= k


def test_topk_wave_by_wave_selection_equals_the_block_wide_bisection(ops, tune):
    """Round 4: top_k <= 64 finds the boundary per wave on ballots and merges 4 x 64 candidates after one barrier (sampler.hip).  Integer
    counts both ways, so ids and pre-draw probabilities must be IDENTICAL to the block-wide bisection (switch sampler_wave_select = 0):
    random rows, rows with ties at and around the boundary, a vocabulary smaller than one wave's share, k = 1 + top-p, k = 64, and the
    nucleus cut taken from the candidate list."""
    g = torch.Generator().manual_seed(5)
    rows = [torch.randn(5, 8324, generator=g) * 3, torch.randn(3, 700, generator=g), torch.randn(2, 40, generator=g),
            torch.randn(2, 12000, generator=g)]
    tie = torch.full((3, 8324), -2.0)
    tie[0, 1000:1100] = 4.0                                  # 100 entries tied for the top: every k cuts inside the tie
    tie[1, ::256] = 1.0; tie[1, 3] = 2.0                     # one wave's lanes hold all the large entries
    tie[2] = torch.randn(8324, generator=g).round()          # few distinct values: ties everywhere
    rows.append(tie)
    rows.append(torch.randn(3, 8324, generator=g) * 12)       # peaked: the nucleus is a handful of entries, most masses round to 0
    rows.append(torch.zeros(2, 8324))                         # flat: every entry in one histogram bucket (falls back to the block-wide passes)
    rows.append(torch.randn(2, 8324, generator=g) * 0.05)     # nearly flat: a few buckets, hundreds of candidates each
    rows.append(torch.cat([torch.full((1, 300), float("-inf")), torch.randn(1, 200, generator=g)], 1))   # -inf entries
    for lg in rows:
        V = lg.shape[1]
        for k, pth, temp in [(1, 0.9, 1.0), (2, None, 1.0), (7, 0.3, 0.7), (50, None, 1.0), (50, 0.9, 1.3), (63, 0.5, 1.0), (64, None, 1.0),
                             (64, 0.999, 1.0), (65, None, 1.0),
                             # the nucleus cut WITHOUT a small top-k: candidates from the mass histogram (sampler.hip, "top-p without a top-k")
                             (None, 0.9, 1.0), (None, 0.3, 0.7), (None, 0.999, 1.0), (None, 0.01, 1.0), (None, 0.5, 0.05), (200, 0.5, 1.0),
                             (100, 0.95, 1.3), (65, 0.9, 1.0)]:
            if k is not None and k >= V:
                continue
            out = []
            for sw in (0, 1):
                tune("sampler_wave_select", sw)
                ids, probs = ops.sample(lg.cuda(), temp, k, pth, seed=9, step=4, want_probs=True)
                out.append((ids.cpu(), probs.cpu()))
            assert torch.equal(out[0][0], out[1][0]), (V, k, pth)
            assert torch.equal(out[0][1], out[1][1]), (V, k, pth)
            kept = (out[1][1] > 0).sum(1)
            assert bool((kept >= 1).all()) and (k is None or bool((kept <= k).all()))
            if pth is None and bool(torch.isfinite(lg).all()):
                assert bool((kept == k).all())


def test_sampler_distribution(ops):
    """Distributional check of the multinomial draw (the reference's torch.multinomial stream
    cannot be reproduced; chi-square style bound on empirical frequencies)."""
    logits = torch.log(torch.tensor([[0.5, 0.25, 0.125, 0.0625, 0.0625]])).repeat(4096, 1)
    counts = torch.zeros(5)
    for step in range(4):
        ids = ops.sample(logits.cuda(), 1.0, None, None, seed=123, step=step).cpu().long()
        counts += torch.bincount(ids, minlength=5).float()
    freq = counts / counts.sum()
    np.testing.assert_allclose(freq.numpy(), [0.5, 0.25, 0.125, 0.0625, 0.0625], atol=0.015)
    a = ops.sample(logits.cuda(), 1.0, None, None, seed=1, step=0)
    b = ops.sample(logits.cuda(), 1.0, None, None, seed=1, step=0)
    c = ops.sample(logits.cuda(), 1.0, None, None, seed=2, step=0)
    assert torch.equal(a, b) and not torch.equal(a, c)


def tile_stats(x):
    """(mean, M2) of every 16-column tile of x [M, K] -> [M, K/16, 2] (what a residual epilogue leaves)."""
    t = x.double().reshape(x.shape[0], -1, 16)
    mean = t.mean(-1)
    return torch.stack([mean, ((t - mean[..., None]) ** 2).sum(-1)], -1).float()


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(64, 1536, 512), (37, 2048, 512), (1, 512, 256), (64, 2304, 768), (16, 3072, 768), (64, 4096, 1024)])
def test_skinny_gemm_ln_gelu(ops, M, N, K):
    """LayerNorm prologue (from 16-column partial statistics) + GEMM + bias + erf-GELU vs fp64, through the
    fragment-ordered layouts (api_cache.py:60-62 and :73)."""
    x = rnd(M, K, seed=1) + 0.3
    w = rnd(N, K, seed=2, scale=K ** -0.5)
    b, g, be = rnd(N, seed=3), 1 + 0.1 * rnd(K, seed=4), 0.1 * rnd(K, seed=5)
    dev = lambda t: t.cuda()
    got = ops.skinny(dev(x), dev(w), dev(b), act=1, ln=(dev(g), dev(be), dev(tile_stats(x)))).cpu()
    xd = x.double()
    xn = (xd - xd.mean(-1, keepdim=True)) / torch.sqrt(xd.var(-1, unbiased=False, keepdim=True) + 1e-5) * g.double() + be.double()
    ref = torch.nn.functional.gelu(xn @ w.double().T + b.double())
    assert (got.double() - ref).abs().max().item() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(64, 512, 512), (5, 512, 2048), (64, 768, 3072), (33, 1024, 1024), (64, 256, 4096)])
def test_skinny_gemm_residual_stats(ops, M, N, K):
    """GEMM + bias + residual in place, and the per-tile (mean, M2) statistics the next LayerNorm merges."""
    a = rnd(M, K, seed=6)
    w = rnd(N, K, seed=7, scale=K ** -0.5)
    b, res = rnd(N, seed=8), rnd(M, N, seed=9)
    got, stats = ops.skinny(a.cuda(), w.cuda(), b.cuda(), residual=res.cuda())
    ref = res.double() + a.double() @ w.double().T + b.double()
    assert (got.cpu().double() - ref).abs().max().item() < 2e-5
    assert (stats.cpu() - tile_stats(ref.float())).abs().max().item() < 1e-4


@pytest.mark.gpu
def test_weight_and_row_tiling_round_trip(ops):
    x = rnd(37, 768, seed=10).cuda()
    assert torch.equal(ops.untile_rows(ops.tile_rows(x), 37, 768), x)
    w = rnd(100, 64, seed=11).cuda()   # 100 rows -> padded to 128
    t = ops.tile_weights(w).cpu()
    assert t.numel() == 128 * 64
    # block (tile 1, chunk 1), h = 1, lane = 16 * 2 + 3  <->  W[16 + 3][32 + 8 * 2 + 4 .. + 3]
    blk = t.view(8, 2, 2, 64, 4)
    assert torch.equal(blk[1, 1, 1, 35], w.cpu()[19, 52:56])
    assert torch.equal(blk[6, 0, 0, 5], torch.zeros(4))   # row 101 is padding


@pytest.mark.gpu
@pytest.mark.parametrize("offset,tol", [(0.0, 2e-5), (10.0, 1e-4), (100.0, 1e-3)])
def test_skinny_gemm_folded_layernorm_with_row_offset(ops, offset, tol):
    """The folded LayerNorm computes rstd * (x @ (gamma*W)^T - mean * c1) + c2: the subtraction cancels when a row's
    mean is large against its spread.  Rows with a DC offset of `offset` standard deviations must stay within the
    north-star logit tolerance (1e-3) -- here: the error bound grows linearly with the offset, as fp32 would predict."""
    M, N, K = 64, 1536, 512
    x = rnd(M, K, seed=21) * 1.7 + offset          # uniform(-1.7, 1.7): sigma ~ 1
    w = rnd(N, K, seed=22, scale=K ** -0.5)
    b, g, be = rnd(N, seed=23), 1 + 0.1 * rnd(K, seed=24), 0.1 * rnd(K, seed=25)
    got = ops.skinny(x.cuda(), w.cuda(), b.cuda(), act=0, ln=(g.cuda(), be.cuda(), tile_stats(x).cuda())).cpu()
    xd = x.double()
    xn = (xd - xd.mean(-1, keepdim=True)) / torch.sqrt(xd.var(-1, unbiased=False, keepdim=True) + 1e-5) * g.double() + be.double()
    ref = xn @ w.double().T + b.double()
    assert (got.double() - ref).abs().max().item() < tol


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(128, 1536, 512), (200, 512, 2048), (256, 2048, 512)])
def test_skinny_gemm_more_than_64_rows(ops, M, N, K):
    """Decode batches of up to MGEA_FUSED_MAX_ROWS (512) rows stay on the fused path: the k-tiled buffers continue in 64-row groups."""
    x = rnd(M, K, seed=31) + 0.1
    w = rnd(N, K, seed=32, scale=K ** -0.5)
    b = rnd(N, seed=33)
    if K <= 1024:
        g, be = 1 + 0.1 * rnd(K, seed=34), 0.1 * rnd(K, seed=35)
        got = ops.skinny(x.cuda(), w.cuda(), b.cuda(), act=1, ln=(g.cuda(), be.cuda(), tile_stats(x).cuda())).cpu()
        xd = x.double()
        xn = (xd - xd.mean(-1, keepdim=True)) / torch.sqrt(xd.var(-1, unbiased=False, keepdim=True) + 1e-5) * g.double() + be.double()
        ref = torch.nn.functional.gelu(xn @ w.double().T + b.double())
        assert (got.double() - ref).abs().max().item() < 2e-5
    else:
        res = rnd(M, N, seed=36)
        got, stats = ops.skinny(x.cuda(), w.cuda(), b.cuda(), residual=res.cuda())
        ref = res.double() + x.double() @ w.double().T + b.double()
        assert (got.cpu().double() - ref).abs().max().item() < 2e-5
        assert (stats.cpu() - tile_stats(ref.float())).abs().max().item() < 1e-4
    y = rnd(M, 768, seed=37).cuda()
    assert torch.equal(ops.untile_rows(ops.tile_rows(y), M, 768), y)


@pytest.mark.parametrize("M,N,K", [(64, 8324, 512), (3, 8324, 512), (33, 8324, 768), (64, 5008, 256), (17, 12401, 512), (48, 4100, 512),
                                   (64, 300, 512), (2, 8324, 512)])
def test_lm_head_balanced_one_round_kernel_equals_the_generic_kernel_bitwise(ops, M, N, K, tune):
    """csrc/head_gemm.hip (round 4): the decode-step head as ONE workgroup per CU with equal unit counts.  Same K split over the
    8 waves, same MFMA chain per output element, same wave-order sum as gemm_skinny_kernel: logits BITWISE equal to the generic kernel's
    (switch head_balanced = 0) for every shape the new kernel takes -- 1, 2 and 3 column tiles per workgroup (N = 5008 / 8324 / 12401),
    K of 1, 2 and 3 chunks per wave, ragged row counts, N not a multiple of 16 or 4 -- and within fp32 summation noise of fp64;
    the greedy (max, argmax) partials merge to torch.argmax with the LOWEST index on exact ties (two duplicated weight rows carry the
    row maximum).  N = 300 and M = 2 are shapes it must decline (the generic kernel / the dot-product path run)."""
    a = rnd(M, K, seed=71)
    w = rnd(N, K, seed=72, scale=K ** -0.5)
    b = rnd(N, seed=73)
    hi = [5, N // 2 + 3, N - 1]
    w[hi[1]] = w[hi[0]]; w[hi[2]] = w[hi[0]]
    b[hi[1]] = b[hi[0]]; b[hi[2]] = b[hi[0]]
    want = (a.double() @ w.double().t() + b.double())
    outs = {}
    for sw in (1, 0):
        tune("head_balanced", sw)
        lg, am, P = ops.head(a.cuda(), w.cuda(), b.cuda())
        outs[sw] = (lg.cpu(), am.cpu(), P)
    takes = 3 <= M <= 64 and K in (256, 512, 768) and N // 16 >= 256
    assert (outs[1][2] != outs[0][2]) == takes, (outs[1][2], outs[0][2])
    if takes:
        assert outs[1][2] % 8 == 0 and outs[1][2] >= 8            # one partial per workgroup = per CU
    assert torch.equal(outs[1][0], outs[0][0]), "balanced head kernel and generic kernel differ bitwise"
    assert float((outs[1][0].double() - want).abs().max()) < 2e-5
    for sw in (1, 0):
        lg, am, _ = outs[sw]
        assert am.tolist() == [int((lg[r] == lg[r].max()).nonzero()[0]) for r in range(M)]     # first index of the row maximum
    # exact ties: make the three duplicated columns every row's maximum, the partial merge must return the first of them
    b2 = b.clone()
    b2[hi] += 100.0
    for sw in (1, 0):
        tune("head_balanced", sw)
        lg, am, _ = ops.head(a.cuda(), w.cuda(), b2.cuda(), want_logits=(sw == 1))
        assert am.cpu().tolist() == [hi[0]] * M
