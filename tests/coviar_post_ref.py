"""Test helpers for the post-decode MV / residual extraction: synthetic AVMotionVector lists, the C
oracle (oracle/coviar_post_ref.c) behind ctypes, an independent pure-Python transcription of the same reference lines for
small cases, and a driver that calls the oracle frame by frame the way the reference's decode_video does
(code/dmcnet/data_loader/coviar_data_loader.c:273-375)."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MV, RESIDUAL = 1, 2

AVMV40 = np.dtype({"names": ["source", "w", "h", "src_x", "src_y", "dst_x", "dst_y", "flags", "motion_x", "motion_y", "motion_scale"],
                   "formats": ["<i4", "u1", "u1", "<i2", "<i2", "<i2", "<i2", "<u8", "<i4", "<i4", "<u2"],
                   "offsets": [0, 4, 5, 6, 8, 10, 12, 16, 24, 28, 32], "itemsize": 40})
AVMV24 = np.dtype({"names": ["source", "w", "h", "src_x", "src_y", "dst_x", "dst_y", "flags"],
                   "formats": ["<i4", "u1", "u1", "<i2", "<i2", "<i2", "<i2", "<u8"],
                   "offsets": [0, 4, 5, 6, 8, 10, 12, 16], "itemsize": 24})

_LIB = None


def oracle_lib():
    global _LIB
    if _LIB is None:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
        _LIB = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libcoviar_post_ref.so"))
    return _LIB


def _p(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else ctypes.c_void_p(0)


def synthetic_mvs(rs, H, W, dtype=AVMV40, max_disp=24, extra=40, zero_frac=0.15):
    """A P-frame's vector list: one 16x16 vector per macroblock (some split into four 8x8), displacement up to
    +-max_disp so that blocks near the border leave the frame, a share of zero-displacement vectors, and `extra`
    vectors of odd sizes at random places that overlap the grid (later ones must win)."""
    recs = []
    for by in range((H + 15) // 16):
        for bx in range((W + 15) // 16):
            if rs.rand() < 0.2:
                subs = [(bx * 16 + 4 + 8 * i, by * 16 + 4 + 8 * j, 8, 8) for j in range(2) for i in range(2)]
            else:
                subs = [(bx * 16 + 8, by * 16 + 8, 16, 16)]
            for dx, dy, w, h in subs:
                if rs.rand() < zero_frac:
                    vx = vy = 0
                else:
                    vx, vy = rs.randint(-max_disp, max_disp + 1, 2)
                recs.append((-1, w, h, dx - vx, dy - vy, dx, dy))
    for _ in range(extra):
        w, h = rs.choice([3, 5, 7, 8, 15, 16, 31]), rs.choice([2, 4, 7, 8, 16, 17])
        dx, dy = rs.randint(-8, W + 8), rs.randint(-8, H + 8)
        vx, vy = rs.randint(-max_disp, max_disp + 1, 2)
        recs.append((-1, w, h, dx - vx, dy - vy, dx, dy))
    out = np.zeros(len(recs), dtype)
    for k, name in enumerate(["source", "w", "h", "src_x", "src_y", "dst_x", "dst_y"]):
        out[name] = [r[k] for r in recs]
    out["flags"] = rs.randint(0, 2, len(recs))          # bytes the extraction must ignore
    if "motion_x" in dtype.names:
        out["motion_x"] = rs.randint(-99, 99, len(recs))
    return out


def c_call(mvs, bgr, mv_arr, res_arr, cur_pos, accumulate, representation, accu_src, accu_old, W, H, pos_target):
    """One call of the oracle's create_and_load_mv_residual restatement; arrays are modified in place."""
    mvs = np.ascontiguousarray(mvs)
    return oracle_lib().cpr_mv_residual(_p(mvs), mvs.dtype.itemsize, mvs.shape[0], _p(bgr), _p(mv_arr), _p(res_arr), cur_pos,
                                        int(accumulate), representation, _p(accu_src), _p(accu_old), W, H, pos_target)


def c_accu_init(H, W):
    accu = np.empty((W, H, 2), np.int32)
    oracle_lib().cpr_accu_init(_p(accu), H, W)
    return accu


def decode_video_policy(frames, representation, accumulate, H, W, call=c_call):
    """What coviar.load returns, given per decoded frame (cur_pos 0 .. pos_target) its side data (or None) and picture:
    the frame-by-frame policy of decode_video, :283-375, around the oracle."""
    pos_target = len(frames) - 1
    bgr = np.zeros((2, H, W, 3), np.uint8)                       # :284-291
    mv_arr = np.zeros((H, W, 2), np.int32)                       # :292-309
    res_arr = np.zeros((H, W, 3), np.int32) if representation == RESIDUAL else None
    accu_src = accu_old = None
    if accumulate:                                               # :306-319
        accu_old = c_accu_init(H, W)
        accu_src = accu_old.copy()
    for cur_pos, (sd, pic) in enumerate(frames):
        if ((cur_pos == 0 and accumulate and representation == RESIDUAL) or
                (cur_pos == pos_target - 1 and not accumulate and representation == RESIDUAL) or cur_pos == pos_target):   # :346-351
            if pic is not None:
                bgr[1 if cur_pos == pos_target else 0] = pic    # :59-66
        if sd is not None and (accumulate or cur_pos == pos_target):          # :362-364
            call(sd, bgr, mv_arr, res_arr, cur_pos, accumulate, representation, accu_src, accu_old, W, H, pos_target)
    return mv_arr if representation == MV else res_arr


def py_call(mvs, bgr, mv_arr, res_arr, cur_pos, accumulate, representation, accu_src, accu_old, W, H, pos_target):
    """Independent transcription of :71-177 in pure Python (small cases only): a second pair of eyes for the C oracle."""
    for m in mvs:
        w, h = int(m["w"]), int(m["h"])
        sx0, sy0, dx0, dy0 = int(m["src_x"]), int(m["src_y"]), int(m["dst_x"]), int(m["dst_y"])
        if dx0 - sx0 == 0 and dy0 - sy0 == 0:
            continue
        for xs in range(-(w // 2), w // 2):                     # C: -1 * w / 2 truncates towards zero = -(w / 2)
            for ys in range(-(h // 2), h // 2):
                pdx, pdy, psx, psy = dx0 + xs, dy0 + ys, sx0 + xs, sy0 + ys
                if 0 <= pdy < H and 0 <= pdx < W and 0 <= psy < H and 0 <= psx < W:
                    if accumulate:
                        accu_src[pdx, pdy, :] = accu_old[psx, psy, :]
                    else:
                        mv_arr[pdy, pdx, 0] = dx0 - sx0
                        mv_arr[pdy, pdx, 1] = dy0 - sy0
    if accumulate:
        accu_old[...] = accu_src
    if cur_pos > 0:
        if accumulate and representation == MV and cur_pos == pos_target:
            for x in range(W):
                for y in range(H):
                    mv_arr[y, x, 0] = x - accu_src[x, y, 0]
                    mv_arr[y, x, 1] = y - accu_src[x, y, 1]
        if representation == RESIDUAL and cur_pos == pos_target:
            for y in range(H):
                for x in range(W):
                    if accumulate:
                        sx, sy = accu_src[x, y]
                    else:
                        sx, sy = x - mv_arr[y, x, 0], y - mv_arr[y, x, 1]
                    res_arr[y, x, :] = bgr[1, y, x].astype(np.int32) - bgr[0, sy, sx].astype(np.int32)
    return 0


def synthetic_gop(rs, H, W, pos_target, dtype=AVMV40, iframe_has_sd=False, target_has_sd=True, **kw):
    """Frames cur_pos 0 .. pos_target of one GOP: (side data or None, picture)."""
    frames = []
    for cur_pos in range(pos_target + 1):
        has = (cur_pos > 0 or iframe_has_sd) and (cur_pos < pos_target or target_has_sd)
        sd = synthetic_mvs(rs, H, W, dtype, **kw) if has else None
        frames.append((sd, rs.randint(0, 256, (H, W, 3)).astype(np.uint8)))
    return frames
