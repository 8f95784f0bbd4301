"""CPU: the C oracle of the post-decode MV / residual extraction (oracle/coviar_post_ref.c) against an independent
pure-Python transcription of the same reference lines (code/dmcnet/data_loader/coviar_data_loader.c:71-177), and hand-made
known answers for the properties the reference's loop order implies.  PARITY UNPINNED for this row: the reference's C file
cannot be built here (FFmpeg absent), see the oracle's header."""
import numpy as np
import pytest

from tests import coviar_post_ref as R


def _mk(recs, dtype=R.AVMV40):
    a = np.zeros(len(recs), dtype)
    for k, name in enumerate(["source", "w", "h", "src_x", "src_y", "dst_x", "dst_y"]):
        a[name] = [r[k] for r in recs]
    return a


def test_later_vector_wins_and_zero_displacement_is_skipped():
    H, W = 12, 20
    mv = np.zeros((H, W, 2), np.int32)
    recs = [(-1, 8, 8, 5, 4, 8, 6),        # writes (3, 2) over x 4..11, y 2..9
            (-1, 8, 8, 9, 6, 10, 6),       # overlaps it with (1, 0) over x 6..13
            (-1, 16, 16, 8, 6, 8, 6)]      # zero displacement: must not clear anything
    assert R.c_call(_mk(recs), None, mv, None, 3, 0, R.MV, None, None, W, H, 3) == 0
    assert (mv[2:10, 4:6] == (3, 2)).all() and (mv[2:10, 6:14] == (1, 0)).all()
    assert (mv[:2] == 0).all() and (mv[10:] == 0).all() and (mv[:, :4] == 0).all() and (mv[:, 14:] == 0).all()


def test_source_outside_frame_blocks_the_pixel_and_odd_sizes_round_towards_zero():
    H, W = 8, 8
    mv = np.zeros((H, W, 2), np.int32)
    # destination columns 0..3 have sources -2..1: only columns 2, 3 pass; w = 5 covers 4 columns (-2 .. 1)
    assert R.c_call(_mk([(-1, 5, 2, 0, 1, 2, 1)]), None, mv, None, 1, 0, R.MV, None, None, W, H, 1) == 0
    assert (mv[0:2, 2:4] == (2, 0)).all() and mv[:, :2].sum() == 0 and mv[:, 4:].sum() == 0 and mv[2:].sum() == 0
    assert R.c_call(_mk([(7, 5, 2, 0, 1, 2, 1)]), None, mv, None, 1, 0, R.MV, None, None, W, H, 1) == 1   # source != -1 counted


@pytest.mark.parametrize("dtype", [R.AVMV40, R.AVMV24])
@pytest.mark.parametrize("representation", [R.MV, R.RESIDUAL])
@pytest.mark.parametrize("accumulate", [0, 1])
@pytest.mark.parametrize("pos_target", [0, 1, 3])
def test_c_oracle_equals_python_transcription(dtype, representation, accumulate, pos_target):
    rs = np.random.RandomState(100 * pos_target + 10 * accumulate + representation)
    H, W = 24, 40
    frames = R.synthetic_gop(rs, H, W, pos_target, dtype, iframe_has_sd=bool(pos_target == 0), max_disp=9, extra=12)
    got = R.decode_video_policy(frames, representation, accumulate, H, W, call=R.c_call)
    ref = R.decode_video_policy(frames, representation, accumulate, H, W, call=R.py_call)
    assert got.dtype == np.int32 and np.array_equal(got, ref)
    if pos_target > 0 and representation == R.MV:
        assert np.abs(got).max() > 0


def test_accumulated_mv_is_the_composition_of_the_frames():
    """Two frames that each shift everything by a constant: accumulated MV = the sum (in the interior)."""
    H, W = 32, 48
    def shift(vx, vy):
        return _mk([(-1, 16, 16, bx * 16 + 8 - vx, by * 16 + 8 - vy, bx * 16 + 8, by * 16 + 8)
                    for by in range(H // 16) for bx in range(W // 16)])
    pic = np.zeros((H, W, 3), np.uint8)
    frames = [(None, pic), (shift(2, 1), pic), (shift(3, -1), pic)]
    mv = R.decode_video_policy(frames, R.MV, 1, H, W)
    assert (mv[8:24, 8:40] == (5, 0)).all()
    assert (R.decode_video_policy(frames, R.MV, 0, H, W)[8:24, 8:40] == (3, -1)).all()
