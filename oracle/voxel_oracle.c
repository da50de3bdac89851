/* Oracle (TEST INFRASTRUCTURE ONLY): scalar C restatement of the reference voxelizers.
 *   voxelgrid_trilinear_c : VoxelGrid.convert            DSEC/dataset/representations.py:15-54
 *   dsec_event_tensor_c   : fixed-count branch + crop      DSEC/dataset/sequence_ov.py:154-165,204-223,281-307
 *   voxelgrid_nearest_i64_c : generate_voxel_grid           datasets/data_util.py:51-117 (int64 events)
 * Same float32/float64 operation order as the NumPy oracle (oracle/events.py); one difference from the
 * reference's 8 put_() passes: contributions are accumulated event by event (corner loop innermost), so
 * sums can differ from the 8-pass order in the last ulp.  Build: make -C oracle  (gcc -O2 -ffp-contract=off). */
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline int trunc_i(float v) { return (v != v) ? INT32_MIN : (v >= 2147483648.0f || v < -2147483648.0f) ? INT32_MIN : (int)v; }

void voxelgrid_trilinear_c(const float* x, const float* y, const float* p, const float* t, int64_t n, int C, int H, int W,
                           int count_mode, float* grid /* C*H*W, zeroed by caller */) {
    if (n <= 0) return;
    const float t0 = t[0], denom = t[n - 1] - t[0];
    for (int64_t i = 0; i < n; ++i) {
        const float tn = ((float)(C - 1) * (t[i] - t0)) / denom;
        const int x0 = trunc_i(x[i]), y0 = trunc_i(y[i]), tt0 = trunc_i(tn);
        const float value = 2.0f * p[i] - 1.0f;
        for (int dx = 0; dx < 2; ++dx)
            for (int dy = 0; dy < 2; ++dy)
                for (int dt = 0; dt < 2; ++dt) {
                    const int64_t xl = (int64_t)x0 + dx, yl = (int64_t)y0 + dy, tl = (int64_t)tt0 + dt;
                    if (xl < 0 || xl >= W || yl < 0 || yl >= H || tl < 0 || tl >= C) continue;
                    float w = value * (1.0f - fabsf((float)xl - x[i]));
                    w = w * (1.0f - fabsf((float)yl - y[i]));
                    w = w * (1.0f - fabsf((float)tl - tn));
                    grid[(int64_t)H * W * tl + (int64_t)W * yl + xl] += count_mode ? 1.0f : w;
                }
    }
}

void dsec_event_tensor_c(const uint16_t* xr, const uint16_t* yr, const int64_t* t_us, const uint8_t* pr, int64_t n_total,
                         const float* rectify_map /* H*W*2 */, int nwin, int C, int H, int W, int crop_rows, int count_mode,
                         float* out /* nwin*C x (H-crop) x W */, float* scratch /* C*H*W + 4*n floats */) {
    const int64_t n = n_total / nwin;
    const int Hout = H - crop_rows;
    float* grid = scratch;
    float* fx = scratch + (int64_t)C * H * W;
    float *fy = fx + n, *fp = fy + n, *ft = fp + n;
    for (int w = 0; w < nwin; ++w) {
        const int64_t s = (int64_t)w * n;
        memset(grid, 0, sizeof(float) * (size_t)C * H * W);
        if (n > 0) {
            const float dlast = (float)(double)(t_us[s + n - 1] - t_us[s]);
            for (int64_t i = 0; i < n; ++i) {
                const float* m = rectify_map + ((int64_t)yr[s + i] * W + xr[s + i]) * 2;
                fx[i] = m[0]; fy[i] = m[1]; fp[i] = (float)pr[s + i];
                ft[i] = (float)(double)(t_us[s + i] - t_us[s]) / dlast;
            }
            voxelgrid_trilinear_c(fx, fy, fp, ft, n, C, H, W, count_mode, grid);
        }
        for (int c = 0; c < C; ++c)
            memcpy(out + ((int64_t)(w * C + c) * Hout) * W, grid + (int64_t)c * H * W, sizeof(float) * (size_t)Hout * W);
    }
}

void voxelgrid_nearest_i64_c(const int64_t* ev /* n x 4: x y t p */, int64_t n, int bins, int H, int W, int separate_pol,
                             int count_mode, float* out /* (separate ? 2 : 1)*bins*H*W */, float* scratch /* 2*bins*H*W */) {
    float* pos = scratch;
    float* neg = scratch + (int64_t)bins * H * W;
    memset(scratch, 0, sizeof(float) * 2 * (size_t)bins * H * W);
    if (n > 0) {
        const int64_t first = ev[2];
        int64_t d = ev[(n - 1) * 4 + 2] - first;
        const double deltaT = d == 0 ? 1.0 : (double)d;
        for (int pass = 0; pass < 4; ++pass)            /* same 4 add.at passes as the reference (order matters in fp32) */
            for (int64_t i = 0; i < n; ++i) {
                const int64_t xs = ev[i * 4], ys = ev[i * 4 + 1];
                int64_t pol = ev[i * 4 + 3];
                if (pol == 0) pol = -1;
                const double ts = (double)((bins - 1) * (ev[i * 4 + 2] - first)) / deltaT;
                const int64_t tis = (int64_t)ts;
                const double dts = ts - (double)tis;
                const double ap = (double)(pol < 0 ? -pol : pol);
                if (!(xs < W && xs >= 0 && ys < H && ys >= 0 && ts >= 0 && ts < bins)) continue;
                const int is_pos = pol == 1;
                float* g = (pass < 2) ? pos : neg;
                if ((pass < 2) != is_pos) continue;
                const int right = pass & 1;
                const int64_t tb = tis + right;
                if (!(tb < bins)) continue;
                const double v = count_mode ? 1.0 : (right ? ap * dts : ap * (1.0 - dts));
                float* cell = &g[xs + ys * W + tb * (int64_t)W * H];
                *cell = (float)((double)*cell + v);
            }
    }
    const int64_t plane = (int64_t)bins * H * W;
    if (separate_pol) memcpy(out, scratch, sizeof(float) * 2 * (size_t)plane);
    else for (int64_t i = 0; i < plane; ++i) out[i] = pos[i] - neg[i];
}
