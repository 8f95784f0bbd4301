"""GPU parity of the post-decode MV / residual extraction (csrc/coviar_post.hip through the C ABI) against the C oracle
(oracle/coviar_post_ref.c): bit-exact, integer work.  Reference: code/dmcnet/data_loader/coviar_data_loader.c:71-175."""
import numpy as np
import pytest
import torch

from tests import coviar_post_ref as R

pytestmark = pytest.mark.gpu


def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1) if a.dtype.names else np.ascontiguousarray(a))
    return t.cuda()


def _lib():
    from dmcnet_amd import _lib
    return _lib


def _stream():
    from dmcnet_amd.ops import _stream
    return _stream()


@pytest.mark.parametrize("dtype", [R.AVMV40, R.AVMV24])
@pytest.mark.parametrize("hw", [(256, 340), (24, 40), (17, 23), (1, 1)])
def test_rasterise_bit_exact(dtype, hw):
    L = _lib()
    lib = L.load()
    H, W = hw
    rs = np.random.RandomState(H * 1000 + W)
    mvs = R.synthetic_mvs(rs, H, W, dtype)
    want = np.zeros((H, W, 2), np.int32)
    want[...] = 7                                     # uncovered pixels keep their value
    R.c_call(mvs, None, want, None, 1, 0, R.MV, None, None, W, H, 1)
    d_mvs = _dev(mvs)
    owner = torch.empty(H * W, dtype=torch.int32, device="cuda")
    out = torch.full((H, W, 2), 7, dtype=torch.int32, device="cuda")
    bad = torch.zeros(1, dtype=torch.int32, device="cuda")
    L.check(lib.dmc_mv_rasterise(L.ptr(d_mvs), dtype.itemsize, len(mvs), L.ptr(owner), L.ptr(out), L.ptr(bad), H, W, _stream()), "rasterise")
    assert np.array_equal(out.cpu().numpy(), want) and int(bad.item()) == 0
    # determinism: the owner map does not depend on the order the atomics land
    out2 = torch.full((H, W, 2), 7, dtype=torch.int32, device="cuda")
    L.check(lib.dmc_mv_rasterise(L.ptr(d_mvs), dtype.itemsize, len(mvs), L.ptr(owner), L.ptr(out2), L._P(0), H, W, _stream()), "rasterise")
    assert torch.equal(out, out2)


def test_bad_source_is_counted_and_arguments_are_checked():
    L = _lib()
    lib = L.load()
    H, W = 64, 64
    mvs = R.synthetic_mvs(np.random.RandomState(3), H, W, extra=0)
    assert len(mvs) >= 16
    mvs["source"][:5] = 1
    owner = torch.empty(H * W, dtype=torch.int32, device="cuda")
    out = torch.zeros((H, W, 2), dtype=torch.int32, device="cuda")
    bad = torch.zeros(1, dtype=torch.int32, device="cuda")
    d = _dev(mvs)
    L.check(lib.dmc_mv_rasterise(L.ptr(d), 40, len(mvs), L.ptr(owner), L.ptr(out), L.ptr(bad), H, W, _stream()), "rasterise")
    assert int(bad.item()) == 5
    assert lib.dmc_mv_rasterise(L.ptr(d), 20, len(mvs), L.ptr(owner), L.ptr(out), L.ptr(bad), H, W, _stream()) == -1
    assert lib.dmc_mv_rasterise(L.ptr(d), 40, len(mvs), L._P(0), L.ptr(out), L.ptr(bad), H, W, _stream()) == -1
    assert lib.dmc_mv_accumulate(L.ptr(d), 40, len(mvs), L.ptr(owner), L.ptr(out), L.ptr(out), L.ptr(bad), H, W, _stream()) == -1
    assert lib.dmc_residual(L.ptr(out), L.ptr(out), L._P(0), L._P(0), L.ptr(out), H, W, _stream()) == -1
    assert lib.dmc_mv_owner_bytes(3, 4, 5) == 240 and lib.dmc_mv_owner_bytes(0, 4, 5) == 0


@pytest.mark.parametrize("hw,n_frames", [((256, 340), 11), ((40, 56), 12), ((17, 23), 3)])
def test_accumulation_chain_step_api_and_batch_api_bit_exact(hw, n_frames):
    """A chain of P-frames through dmc_mv_accumulate (one call per frame, buffers swapped) and through
    dmc_mv_gop_batch (all frames at once, every pixel walked back): both equal the oracle's accumulator, the MV derived
    from it and the residual."""
    L = _lib()
    lib = L.load()
    H, W = hw
    rs = np.random.RandomState(n_frames * 7 + H)
    frames = [R.synthetic_mvs(rs, H, W) for _ in range(n_frames)]
    pics = rs.randint(0, 256, (2, H, W, 3)).astype(np.uint8)
    accu_old = R.c_accu_init(H, W)
    accu_src = accu_old.copy()
    mv_want = np.zeros((H, W, 2), np.int32)
    res_want = np.zeros((H, W, 3), np.int32)
    for t, mvs in enumerate(frames):
        R.c_call(mvs, pics, mv_want, None, t + 1, 1, R.MV, accu_src, accu_old, W, H, n_frames)
    R.c_call(np.zeros(0, R.AVMV40), pics, mv_want.copy(), res_want, n_frames, 1, R.RESIDUAL, accu_src, accu_old, W, H, n_frames)

    owner = torch.empty(H * W, dtype=torch.int32, device="cuda")
    a = torch.empty((W, H, 2), dtype=torch.int32, device="cuda")
    b = torch.empty_like(a)
    L.check(lib.dmc_mv_accu_init(L.ptr(a), H, W, _stream()), "accu_init")
    assert np.array_equal(a.cpu().numpy(), R.c_accu_init(H, W))
    for mvs in frames:
        d = _dev(mvs)
        L.check(lib.dmc_mv_accumulate(L.ptr(d), 40, len(mvs), L.ptr(owner), L.ptr(a), L.ptr(b), L._P(0), H, W, _stream()), "accumulate")
        a, b = b, a
    assert np.array_equal(a.cpu().numpy(), accu_src)
    mv = torch.empty((H, W, 2), dtype=torch.int32, device="cuda")
    res = torch.empty((H, W, 3), dtype=torch.int32, device="cuda")
    d_pics = torch.from_numpy(pics).cuda()
    L.check(lib.dmc_mv_from_accu(L.ptr(a), L.ptr(mv), H, W, _stream()), "from_accu")
    L.check(lib.dmc_residual(L.ptr(d_pics[0]), L.ptr(d_pics[1]), L.ptr(a), L._P(0), L.ptr(res), H, W, _stream()), "residual")
    assert np.array_equal(mv.cpu().numpy(), mv_want) and np.array_equal(res.cpu().numpy(), res_want)

    allmv = torch.cat([_dev(f) for f in frames])       # bytes: np.concatenate would re-pack the padded record dtype
    off = np.cumsum([0] + [len(f) for f in frames]).astype(np.int32)
    d_off = torch.from_numpy(off).cuda()
    d_chain = torch.tensor([0, n_frames], dtype=torch.int32, device="cuda")
    owners = torch.empty(lib.dmc_mv_owner_bytes(n_frames, H, W) // 4, dtype=torch.int32, device="cuda")
    accu2 = torch.empty((W, H, 2), dtype=torch.int32, device="cuda")
    mv2 = torch.empty_like(mv)
    res2 = torch.empty_like(res)
    L.check(lib.dmc_mv_gop_batch(L.ptr(allmv), 40, int(off[-1]), L.ptr(d_off), n_frames, L.ptr(d_chain), 1, L._P(0), L.ptr(owners),
                                 L.ptr(d_pics[0]), L.ptr(d_pics[1]), L.ptr(accu2), L.ptr(mv2), L.ptr(res2), L._P(0), H, W, _stream()), "gop_batch")
    assert np.array_equal(accu2.cpu().numpy(), accu_src)
    assert np.array_equal(mv2.cpu().numpy(), mv_want) and np.array_equal(res2.cpu().numpy(), res_want)


def test_residual_from_mv_plane_bit_exact():
    L = _lib()
    lib = L.load()
    H, W = 64, 96
    rs = np.random.RandomState(5)
    mvs = R.synthetic_mvs(rs, H, W)
    pics = rs.randint(0, 256, (2, H, W, 3)).astype(np.uint8)
    mv_want = np.zeros((H, W, 2), np.int32)
    res_want = np.zeros((H, W, 3), np.int32)
    R.c_call(mvs, pics, mv_want, res_want, 4, 0, R.RESIDUAL, None, None, W, H, 4)
    d_pics = torch.from_numpy(pics).cuda()
    d_mv = torch.from_numpy(mv_want).cuda()
    res = torch.empty((H, W, 3), dtype=torch.int32, device="cuda")
    L.check(lib.dmc_residual(L.ptr(d_pics[0]), L.ptr(d_pics[1]), L._P(0), L.ptr(d_mv), L.ptr(res), H, W, _stream()), "residual")
    assert np.array_equal(res.cpu().numpy(), res_want)


@pytest.mark.parametrize("dtype", [R.AVMV40, R.AVMV24])
@pytest.mark.parametrize("representation", [R.MV, R.RESIDUAL])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_host_mirror_equals_decode_video_policy(dtype, representation, accumulate):
    """coviar_post.extract_batch (what coviar.load returns, per sample) for a batch that mixes every gate of the
    reference's decode_video: pos_target 0 with and without side data, I-frame without side data, a target frame
    without side data, short and full-GOP chains."""
    from dmcnet_amd import coviar_post
    H, W = 48, 64
    rs = np.random.RandomState(31 + 2 * accumulate + representation)
    samples = [R.synthetic_gop(rs, H, W, 0, dtype), R.synthetic_gop(rs, H, W, 0, dtype, iframe_has_sd=True),
               R.synthetic_gop(rs, H, W, 1, dtype), R.synthetic_gop(rs, H, W, 5, dtype),
               R.synthetic_gop(rs, H, W, 4, dtype, target_has_sd=False), R.synthetic_gop(rs, H, W, 11, dtype, iframe_has_sd=True)]
    got = coviar_post.extract_batch(samples, representation, accumulate).cpu().numpy()
    for i, frames in enumerate(samples):
        want = R.decode_video_policy(frames, representation, accumulate, H, W)
        assert np.array_equal(got[i], want), i
    one = coviar_post.load_post_decode(samples[3], representation, accumulate).cpu().numpy()
    assert np.array_equal(one, got[3])


@pytest.mark.slow
def test_full_batch_120_chains_340x256():
    """BASELINE-size batch (40 clips x 3 segments, 340 x 256 MPEG-4 frames, chains of 1 .. 11 P-frames): the batch
    call equals the oracle on sampled chains, is bit-identical when repeated, and a chain's result does not depend on
    what else is in the batch."""
    from dmcnet_amd import coviar_post
    H, W = 256, 340
    rs = np.random.RandomState(77)
    samples = [R.synthetic_gop(rs, H, W, 1 + (i % 11), extra=10) for i in range(120)]
    got = coviar_post.extract_batch(samples, R.MV, 1)
    again = coviar_post.extract_batch(samples, R.MV, 1)
    assert torch.equal(got, again)
    res = coviar_post.extract_batch(samples, R.RESIDUAL, 1)
    for i in (0, 10, 53, 119):
        assert np.array_equal(got[i].cpu().numpy(), R.decode_video_policy(samples[i], R.MV, 1, H, W)), i
        assert np.array_equal(res[i].cpu().numpy(), R.decode_video_policy(samples[i], R.RESIDUAL, 1, H, W)), i
    alone = coviar_post.extract_batch(samples[53:54], R.MV, 1)
    assert torch.equal(alone[0], got[53])
