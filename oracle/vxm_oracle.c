/* TEST INFRASTRUCTURE ONLY — plain-C restatement of the VxmDense hot-path arithmetic.
 *
 * Not linked into, imported by or shipped with the product (voxelmorph_amd).  Used by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as an ATen-independent
 * checker.  Each function cites the reference lines it follows (paths relative to
 * /root/reference).  Build: `make -C oracle` (gcc -O2 -ffp-contract=off: the nearest-neighbour
 * warp must reproduce the reference's fp32 op order exactly).
 *
 * Pinned by tests/test_oracle_golden.py against the npz fixtures under tests/golden (reference outputs).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* voxelmorph/torch/layers.py:32,36-37 + ATen grid_sampler unnormalize (align_corners=True):
 * loc = i + f ; c = 2*(loc/(S-1) - 0.5) ; x = ((c+1)/2)*(S-1).  All fp32, no contraction. */
static float src_coord(int i, float f, int S) {
    volatile float loc = (float)i + f;
    volatile float q = loc / (float)(S - 1);
    volatile float c = 2.0f * (q - 0.5f);
    volatile float h = (c + 1.0f) / 2.0f;
    volatile float x = h * (float)(S - 1);
    return x;
}

/* layers.py:30-48 (3-D): out[b,c,p] = sample(src[b,c], p + flow[b,:,p]); zeros outside.
 * mode 0 = trilinear ('bilinear'), 1 = nearest (round-half-even, nearbyint). */
void orc_warp3d(const float* src, const float* flow, float* out, int B, int C, int D, int H, int W,
                int mode) {
    const long V = (long)D * H * W;
    for (int b = 0; b < B; ++b)
        for (int d = 0; d < D; ++d)
            for (int h = 0; h < H; ++h)
                for (int w = 0; w < W; ++w) {
                    const long p = ((long)d * H + h) * W + w;
                    const float* fl = flow + (long)b * 3 * V;
                    const float z = src_coord(d, fl[p], D);
                    const float y = src_coord(h, fl[V + p], H);
                    const float x = src_coord(w, fl[2 * V + p], W);
                    if (mode == 1) {
                        const long jz = (long)nearbyintf(z), jy = (long)nearbyintf(y), jx = (long)nearbyintf(x);
                        const int ok = jz >= 0 && jz < D && jy >= 0 && jy < H && jx >= 0 && jx < W;
                        for (int c = 0; c < C; ++c)
                            out[((long)b * C + c) * V + p] =
                                ok ? src[((long)b * C + c) * V + (jz * H + jy) * W + jx] : 0.0f;
                        continue;
                    }
                    const float z0 = floorf(z), y0 = floorf(y), x0 = floorf(x);
                    for (int c = 0; c < C; ++c) {
                        const float* s = src + ((long)b * C + c) * V;
                        float acc = 0.0f;
                        for (int k = 0; k < 8; ++k) {
                            const float cz = z0 + (float)((k >> 2) & 1), cy = y0 + (float)((k >> 1) & 1),
                                        cx = x0 + (float)(k & 1);
                            const long jz = (long)cz, jy = (long)cy, jx = (long)cx;
                            if (jz < 0 || jz >= D || jy < 0 || jy >= H || jx < 0 || jx >= W) continue;
                            const float wgt = (1.0f - fabsf(z - cz)) * (1.0f - fabsf(y - cy)) * (1.0f - fabsf(x - cx));
                            acc += s[(jz * H + jy) * W + jx] * wgt;
                        }
                        out[((long)b * C + c) * V + p] = acc;
                    }
                }
}

/* layers.py:64-68: vec *= 1/2^n; n times: vec = vec + warp(vec, vec).  `out` holds the result,
 * `tmp` is scratch of the same size. */
void orc_vecint3d(const float* vec, float* out, float* tmp, int B, int D, int H, int W, int nsteps) {
    const long n = (long)B * 3 * D * H * W;
    const float scale = 1.0f / (float)(1 << nsteps);
    for (long i = 0; i < n; ++i) out[i] = vec[i] * scale;
    for (int s = 0; s < nsteps; ++s) {
        orc_warp3d(out, out, tmp, B, 3, D, H, W, 0);
        for (long i = 0; i < n; ++i) out[i] = out[i] + tmp[i];
    }
}

/* layers.py:85-97 + ATen upsample_trilinear3d(align_corners=True): src = dst*(in-1)/(out-1);
 * factor<1: resize then scale; factor>1: scale then resize. */
void orc_resize3d(const float* x, float* out, int B, int C, int D, int H, int W, int oD, int oH, int oW,
                  float factor) {
    const float sd = oD > 1 ? (float)(D - 1) / (float)(oD - 1) : 0.0f;
    const float sh = oH > 1 ? (float)(H - 1) / (float)(oH - 1) : 0.0f;
    const float sw = oW > 1 ? (float)(W - 1) / (float)(oW - 1) : 0.0f;
    const float pre = factor > 1.0f ? factor : 1.0f, post = factor < 1.0f ? factor : 1.0f;
    for (long bc = 0; bc < (long)B * C; ++bc) {
        const float* s = x + bc * D * H * W;
        float* o = out + bc * (long)oD * oH * oW;
        for (int d = 0; d < oD; ++d) {
            const float fz = sd * (float)d;
            int z0 = (int)fz; if (z0 > D - 1) z0 = D - 1;
            const int z1 = z0 + (z0 < D - 1);
            const float lz1 = fz - (float)z0, lz0 = 1.0f - lz1;
            for (int h = 0; h < oH; ++h) {
                const float fy = sh * (float)h;
                int y0 = (int)fy; if (y0 > H - 1) y0 = H - 1;
                const int y1 = y0 + (y0 < H - 1);
                const float ly1 = fy - (float)y0, ly0 = 1.0f - ly1;
                for (int w = 0; w < oW; ++w) {
                    const float fx = sw * (float)w;
                    int x0 = (int)fx; if (x0 > W - 1) x0 = W - 1;
                    const int x1 = x0 + (x0 < W - 1);
                    const float lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
#define AT(zz, yy, xx) (pre * s[((long)(zz) * H + (yy)) * W + (xx)])
                    const float v =
                        lz0 * (ly0 * (lx0 * AT(z0, y0, x0) + lx1 * AT(z0, y0, x1)) +
                               ly1 * (lx0 * AT(z0, y1, x0) + lx1 * AT(z0, y1, x1))) +
                        lz1 * (ly0 * (lx0 * AT(z1, y0, x0) + lx1 * AT(z1, y0, x1)) +
                               ly1 * (lx0 * AT(z1, y1, x0) + lx1 * AT(z1, y1, x1)));
#undef AT
                    o[((long)d * oH + h) * oW + w] = post * v;
                }
            }
        }
    }
}

/* losses.py:102-135: mean over batch of loss_mult * (1/3) sum_axis mean(|d|^p), fp64 accumulate. */
double orc_grad_loss(const float* y, int B, int C, int D, int H, int W, int l2, double mult) {
    double total = 0.0;
    const long V = (long)D * H * W;
    for (int b = 0; b < B; ++b) {
        double sd = 0, sh = 0, sw = 0;
        for (int c = 0; c < C; ++c) {
            const float* p = y + ((long)b * C + c) * V;
            for (int d = 0; d < D; ++d)
                for (int h = 0; h < H; ++h)
                    for (int w = 0; w < W; ++w) {
                        const long i = ((long)d * H + h) * W + w;
                        if (d + 1 < D) { const float t = p[i + (long)H * W] - p[i]; sd += l2 ? (double)(t * t) : fabs((double)t); }
                        if (h + 1 < H) { const float t = p[i + W] - p[i]; sh += l2 ? (double)(t * t) : fabs((double)t); }
                        if (w + 1 < W) { const float t = p[i + 1] - p[i]; sw += l2 ? (double)(t * t) : fabs((double)t); }
                    }
        }
        const double md = sd / ((double)C * (D - 1) * H * W), mh = sh / ((double)C * D * (H - 1) * W),
                     mw = sw / ((double)C * D * H * (W - 1));
        total += mult * (md + mh + mw) / 3.0;
    }
    return total / B;
}

static void box1d(const double* in, double* out, long n_lines, int len, long stride_line_major,
                  long stride_elem, int r) {
    (void)stride_line_major;
    for (long l = 0; l < n_lines; ++l) {
        const double* a = in + l;      /* caller arranges layout so that lines are indexed by l */
        double* o = out + l;
        for (int i = 0; i < len; ++i) {
            double s = 0;
            for (int k = -r; k <= r; ++k) { const int j = i + k; if (j >= 0 && j < len) s += a[j * stride_elem]; }
            o[i * stride_elem] = s;
        }
    }
}

/* zero-padded win^3 box sum of one [D,H,W] volume in fp64 (losses.py:51-55 use conv3d with a
 * ones filter and padding win/2) */
static void box3d(const double* in, double* out, double* tmp, int D, int H, int W, int win) {
    const int r = win / 2;
    const long V = (long)D * H * W;
    /* along w */
    for (long l = 0; l < (long)D * H; ++l) box1d(in + l * W, tmp + l * W, 1, W, 0, 1, r);
    /* along h */
    for (int d = 0; d < D; ++d)
        for (int w = 0; w < W; ++w) box1d(tmp + (long)d * H * W + w, out + (long)d * H * W + w, 1, H, 0, W, r);
    /* along d */
    for (long l = 0; l < (long)H * W; ++l) box1d(out + l, tmp + l, 1, D, 0, (long)H * W, r);
    memcpy(out, tmp, sizeof(double) * V);
}

/* losses.py:47-67 in fp64 (arbiter): -mean(cross^2 / (I_var*J_var + 1e-5)). */
double orc_ncc_loss(const float* I, const float* J, int B, int D, int H, int W, int win) {
    const long V = (long)D * H * W;
    const double n = (double)win * win * win;
    double* buf = (double*)malloc(sizeof(double) * V * 7);
    double *q = buf, *tmp = buf + V, *S[5] = {buf + 2 * V, buf + 3 * V, buf + 4 * V, buf + 5 * V, buf + 6 * V};
    double total = 0;
    for (int b = 0; b < B; ++b) {
        const float *Ib = I + b * V, *Jb = J + b * V;
        for (int k = 0; k < 5; ++k) {
            for (long i = 0; i < V; ++i) {
                const double a = Ib[i], c = Jb[i];
                q[i] = k == 0 ? a : k == 1 ? c : k == 2 ? a * a : k == 3 ? c * c : a * c;
            }
            box3d(q, S[k], tmp, D, H, W, win);
        }
        for (long i = 0; i < V; ++i) {
            const double Is = S[0][i], Js = S[1][i], I2 = S[2][i], J2 = S[3][i], IJ = S[4][i];
            const double uI = Is / n, uJ = Js / n;
            const double cross = IJ - uJ * Is - uI * Js + uI * uJ * n;
            const double Iv = I2 - 2 * uI * Is + uI * uI * n, Jv = J2 - 2 * uJ * Js + uJ * uJ * n;
            total += cross * cross / (Iv * Jv + 1e-5);
        }
    }
    free(buf);
    return -total / ((double)B * V);
}

/* networks.py:299-305 ConvBlock: y = leaky_relu(conv3d(x, w, b, k=3, pad=1), slope); slope==1
 * gives the bare flow conv (networks.py:211,257).  Direct sum, fp64 accumulate (arbiter). */
void orc_conv3d_k3(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout,
                   int D, int H, int W, float slope) {
    const long V = (long)D * H * W;
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int d = 0; d < D; ++d)
                for (int h = 0; h < H; ++h)
                    for (int wq = 0; wq < W; ++wq) {
                        double acc = bias ? bias[co] : 0.0;
                        for (int ci = 0; ci < Cin; ++ci) {
                            const float* xp = x + ((long)b * Cin + ci) * V;
                            const float* wp = w + ((long)co * Cin + ci) * 27;
                            for (int kd = 0; kd < 3; ++kd) {
                                const int zz = d + kd - 1; if (zz < 0 || zz >= D) continue;
                                for (int kh = 0; kh < 3; ++kh) {
                                    const int yy = h + kh - 1; if (yy < 0 || yy >= H) continue;
                                    for (int kw = 0; kw < 3; ++kw) {
                                        const int xx = wq + kw - 1; if (xx < 0 || xx >= W) continue;
                                        acc += (double)xp[((long)zz * H + yy) * W + xx] * wp[(kd * 3 + kh) * 3 + kw];
                                    }
                                }
                            }
                        }
                        const float v = (float)acc;
                        y[((long)b * Cout + co) * V + ((long)d * H + h) * W + wq] = v > 0 ? v : v * slope;
                    }
}
