/* TEST INFRASTRUCTURE -- CPU oracle for the post-decode MV / residual extraction.
 *
 * A plain-C restatement of what the reference's data loader does with the arrays FFmpeg hands it:
 * code/dmcnet/data_loader/coviar_data_loader.c:71-175 (create_and_load_mv_residual) and :306-319 (accumulator set-up
 * inside decode_video).  Sequential, one vector after the other, so "the later vector wins" is simply program order.
 *
 * PARITY UNPINNED: the reference file needs FFmpeg (libavcodec / libavutil / libswscale) and the CPython + numpy C API
 * to compile; FFmpeg's headers and libraries are absent from this image and writing stand-ins for them would pin
 * nothing, so this restatement could not be checked against a build of the reference.  It is cross-checked against an
 * independent pure-Python transcription of the same lines (tests/test_coviar_post_cpu.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load this.  Third-party layout restated here:
 * AVMotionVector of libavutil/motion_vector.h (FFmpeg's public ABI; the CoViAR loader this file derives from was written
 * against FFmpeg 3.x): int32 source; uint8 w, h; int16 src_x, src_y, dst_x, dst_y; uint64 flags [; int32 motion_x,
 * motion_y; uint16 motion_scale from libavutil 55.63 on].  Records are read at byte offsets with the caller's stride.
 */
#include <stdint.h>
#include <string.h>

typedef struct {
    int source, w, h, src_x, src_y, dst_x, dst_y;
} mv_fields;

static mv_fields read_mv(const uint8_t* base, int stride, long i) {
    const uint8_t* p = base + (size_t)i * (size_t)stride;
    mv_fields m;
    int32_t s32;
    int16_t s16;
    memcpy(&s32, p, 4);      m.source = s32;
    m.w = p[4];
    m.h = p[5];
    memcpy(&s16, p + 6, 2);  m.src_x = s16;
    memcpy(&s16, p + 8, 2);  m.src_y = s16;
    memcpy(&s16, p + 10, 2); m.dst_x = s16;
    memcpy(&s16, p + 12, 2); m.dst_y = s16;
    return m;
}

/* :311-318 */
void cpr_accu_init(int32_t* accu, int H, int W) {
    for (int x = 0; x < W; ++x)
        for (int y = 0; y < H; ++y) {
            accu[(x * H + y) * 2] = x;
            accu[(x * H + y) * 2 + 1] = y;
        }
}

/* One call of create_and_load_mv_residual (:71-177) without the Python objects.
 *   mvs / stride / n_mv : the side data (sd->data, sizeof(*mvs), sd->size / sizeof(*mvs))
 *   bgr                 : uint8 [2][H][W][3] (index 0 = reference frame, 1 = the target frame), may be NULL unless a
 *                         residual is produced
 *   mv_arr, res_arr     : int32 [H][W][2] / [H][W][3]
 *   accu_src, accu_old  : int32 [W][H][2], used when accumulate != 0
 *   representation      : 1 = MV, 2 = RESIDUAL (the reference's #defines)
 * Returns the number of vectors whose source is not -1 (the reference asserts on them, :86). */
int cpr_mv_residual(const uint8_t* mvs, int stride, int n_mv, const uint8_t* bgr, int32_t* mv_arr, int32_t* res_arr, int cur_pos,
                    int accumulate, int representation, int32_t* accu_src, int32_t* accu_old, int W, int H, int pos_target) {
    int bad = 0;
    for (long i = 0; i < n_mv; ++i) {
        const mv_fields m = read_mv(mvs, stride, i);
        if (m.source != -1) ++bad;
        const int vx = m.dst_x - m.src_x, vy = m.dst_y - m.src_y;
        if (vx == 0 && vy == 0) continue;                                   /* :88 */
        for (int ox = (-1 * m.w) / 2; ox < m.w / 2; ++ox)                    /* :91 (C integer division) */
            for (int oy = (-1 * m.h) / 2; oy < m.h / 2; ++oy) {              /* :92 */
                const int dx = m.dst_x + ox, dy = m.dst_y + oy, sx = m.src_x + ox, sy = m.src_y + oy;
                if (!(dy >= 0 && dy < H && dx >= 0 && dx < W && sy >= 0 && sy < H && sx >= 0 && sx < W)) continue;   /* :100-103 */
                if (accumulate) {                                           /* :106-110 */
                    accu_src[(dx * H + dy) * 2] = accu_old[(sx * H + sy) * 2];
                    accu_src[(dx * H + dy) * 2 + 1] = accu_old[(sx * H + sy) * 2 + 1];
                } else {                                                    /* :112-113 */
                    mv_arr[(dy * W + dx) * 2] = vx;
                    mv_arr[(dy * W + dx) * 2 + 1] = vy;
                }
            }
    }
    if (accumulate) memcpy(accu_old, accu_src, (size_t)W * H * 2 * sizeof(int32_t));   /* :125-127 */
    if (cur_pos > 0) {                                                                   /* :128 */
        if (accumulate && representation == 1 && cur_pos == pos_target)                  /* :129-139 */
            for (int x = 0; x < W; ++x)
                for (int y = 0; y < H; ++y) {
                    mv_arr[(y * W + x) * 2] = x - accu_src[(x * H + y) * 2];
                    mv_arr[(y * W + x) * 2 + 1] = y - accu_src[(x * H + y) * 2 + 1];
                }
        if (representation == 2 && cur_pos == pos_target) {                              /* :141-175 */
            const uint8_t* ref = bgr;
            const uint8_t* cur = bgr + (size_t)H * W * 3;
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    int sx, sy;
                    if (accumulate) {
                        sx = accu_src[(x * H + y) * 2];
                        sy = accu_src[(x * H + y) * 2 + 1];
                    } else {
                        sx = x - mv_arr[(y * W + x) * 2];
                        sy = y - mv_arr[(y * W + x) * 2 + 1];
                    }
                    for (int c = 0; c < 3; ++c)
                        res_arr[(y * W + x) * 3 + c] = (int32_t)cur[(y * W + x) * 3 + c] - (int32_t)ref[(sy * W + sx) * 3 + c];
                }
        }
    }
    return bad;
}
