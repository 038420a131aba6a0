// The ResNet stem: 7x7 / stride 2 / pad 3 convolution of the 4-channel (RGB + zero) image to 64 channels (`conv0`, resnet.py:248-249),
// with the input batch-norm (`bn_data`) as a per-channel affine on the real pixels and bn0 + ReLU as the epilogue.
//
// In the implicit-GEMM kernel this layer is bound by operand delivery (every input pixel is fetched again for ~12 of the 49 taps and
// feeds only 64 MACs per fetch: 0.43 ms, the same time a plain GEMM of its shape takes).  Same cure as conv_halo.hip: a persistent
// block owns a 4 x 32 tile of OUTPUT pixels, stages the (2*4+5) x (2*32+5) input halo in LDS once (14 KB, two stages, filled by the four
// producer waves with the input affine applied) and runs all 49 taps from it; K = 49 taps x 4 channels is walked in 25 steps of two taps
// (half-wave 0 takes tap 2s, half-wave 1 tap 2s+1; tap 49 does not exist: zero weights); the weight fragments come straight from L1/L2
// through a 4-step register ring; accumulators are transposed (lane = pixel) so the epilogue moves 16 bytes per access.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 4, TW = 32;                 // output tile
constexpr int HR = 2 * TH + 5, HC = 2 * TW + 5;  // input halo (13 x 69 pixels of 4 floats)
constexpr int HP = HR * HC;
constexpr int NSTEP = 25;                      // ceil(49 / 2) two-tap steps
constexpr int RING = 5, RD = RING - 1;         // 25 % 5 == 0: ring slots stay compile-time constants across tiles
constexpr int NFILL = (HP + 255) / 256;

struct StemK {
    const float* img;        // [B,H,W,4]
    unsigned img_bytes;
    const float* pre_scale;  // [4] or null
    const float* pre_shift;
    const float* W;          // [25][2][64][4]
    const float* scale;      // [64] or null
    const float* shift;
    int act;
    float* out_raw; int raw_ld;
    float* out_act; int act_ld;
    int B, H, Wd, Ho, Wo, tiles_y, tiles_x, ntiles;
};

#define CP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__global__ __launch_bounds__(512, 4) void conv_stem_kernel(const StemK p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][HP][4]
    constexpr unsigned OOB = 0x80000000u;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave >= 4;
    const int tid = threadIdx.x & 255;
    const int lane = tid & 63;
    const int bid = cp::xcd_remap(blockIdx.x, gridDim.x);
    const int my_tiles = (p.ntiles - bid + (int)gridDim.x - 1) / (int)gridDim.x;
    if (my_tiles <= 0) return;

    auto tile_pos = [&](int k, int& n, int& y0, int& x0) {
        int t = bid + k * (int)gridDim.x;
        x0 = (t % p.tiles_x) * TW;
        t /= p.tiles_x;
        y0 = (t % p.tiles_y) * TH;
        n = t / p.tiles_y;
    };

    if (producer) {
        const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc((void*)p.img, 0, p.img_bytes, 0x00020000);
        float4 ps = make_float4(1.f, 1.f, 1.f, 1.f), pb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.pre_scale) {
            ps = *reinterpret_cast<const float4*>(p.pre_scale);
            pb = *reinterpret_cast<const float4*>(p.pre_shift);
        }
        int e_hy[NFILL], e_hx[NFILL];
#pragma unroll
        for (int i = 0; i < NFILL; ++i) {
            const int e = tid + 256 * i;
            e_hy[i] = e < HP ? e / HC : 0x4000;  // 0x4000: never in bounds
            e_hx[i] = e % HC;
        }
        auto fill = [&](int k, int stage) {
            int n, y0, x0;
            tile_pos(k, n, y0, x0);
            float4 v[NFILL];
            bool inb[NFILL];
#pragma unroll
            for (int i = 0; i < NFILL; ++i) {
                const int y = 2 * y0 - 3 + e_hy[i], x = 2 * x0 - 3 + e_hx[i];
                inb[i] = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.Wd;
                v[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsi, inb[i] ? (int)((((unsigned)n * p.H + y) * p.Wd + x) * 16u) : (int)OOB, 0, 0));
            }
            float* h = smem + stage * HP * 4;
#pragma unroll
            for (int i = 0; i < NFILL; ++i) {
                if (e_hy[i] >= 0x4000) continue;
                float4 r = v[i];
                if (inb[i]) {  // the affine applies to real pixels only: padding stays exactly zero
                    r.x = r.x * ps.x + pb.x; r.y = r.y * ps.y + pb.y; r.z = r.z * ps.z + pb.z; r.w = r.w * ps.w + pb.w;
                }
                *reinterpret_cast<float4*>(h + (tid + 256 * i) * 4) = r;
            }
        };
        fill(0, 0);
        CP_BARRIER();
        for (int k = 0; k < my_tiles; ++k) {
            if (k + 1 < my_tiles) fill(k + 1, (k + 1) & 1);   // stage (k+1)&1 was last read for tile k-1
            CP_BARRIER();
        }
        return;
    }

    // ------------------------------------ consumers: wave w owns output row w of the tile ------------------------------------
    const int wy = wave;
    const int lrow = lane & 31, half = lane >> 5;
    const unsigned npix = (unsigned)(p.B * p.Ho * p.Wo);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, NSTEP * 2 * 64 * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_tab_s = __builtin_amdgcn_make_buffer_rsrc((void*)(p.scale ? p.scale : p.W), 0, p.scale ? 256u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_tab_b = __builtin_amdgcn_make_buffer_rsrc((void*)(p.scale ? p.shift : p.W), 0, p.scale ? 256u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_raw = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out_raw ? (void*)p.out_raw : (void*)p.W), 0,
                                                                            p.out_raw ? npix * (unsigned)p.raw_ld * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_act = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out_act ? (void*)p.out_act : (void*)p.W), 0,
                                                                            p.out_act ? npix * (unsigned)p.act_ld * 4u : 0u, 0x00020000);
    auto ldw = [&](int step, int j) -> float4 {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsw, (int)((((unsigned)step * 2 + half) * 64 + j * 32 + lrow) * 16u), 0, 0));
    };
    float4 fb[RING][2];
#pragma unroll
    for (int u = 0; u < RD; ++u)
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[u][j] = ldw(u, j);
    f32x16 acc[2];
    CP_BARRIER();  // halo of tile 0 is in LDS
    for (int k = 0; k < my_tiles; ++k) {
        int n, y0, x0;
        tile_pos(k, n, y0, x0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const float* hb = smem + (k & 1) * HP * 4 + ((2 * wy) * HC + 2 * lrow) * 4;
        float4 fa[2];
        auto read_a = [&](int s, int slot) {
            const int tl = 2 * s, th = (2 * s + 1 < 49) ? 2 * s + 1 : 48;   // tap 49 does not exist: its weights are zero
            const int offl = ((tl / 7) * HC + tl % 7) * 4, offh = ((th / 7) * HC + th % 7) * 4;
            fa[slot] = *reinterpret_cast<const float4*>(hb + (half ? offh : offl));
        };
        read_a(0, 0);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[(s + RD) % RING][j] = ldw((s + RD) % NSTEP, j);   // past the end: the next tile's first steps
            if (s + 1 < NSTEP) read_a(s + 1, (s + 1) & 1);
            const float4 av = fa[s & 1];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 w = fb[s % RING][j];
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, av.x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, av.y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, av.z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, av.w, acc[j], 0, 0, 0);
            }
        }
        // ---- epilogue: lane = pixel; register r of block j = channel j*32 + (r&3) + 8*(r>>2) + 4*half ----
        const int y = y0 + wy, x = x0 + lrow;
        const bool pok = y < p.Ho && x < p.Wo;
        const unsigned pix = (unsigned)((n * p.Ho + y) * p.Wo + x);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float4 sc[4], sh[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int ch = j * 32 + g4 * 8 + half * 4;
                sc[g4] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_tab_s, ch * 4, 0, 0));
                sh[g4] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r_tab_b, ch * 4, 0, 0));
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int ch = j * 32 + g4 * 8 + half * 4;
                float4 v = make_float4(acc[j][g4 * 4 + 0], acc[j][g4 * 4 + 1], acc[j][g4 * 4 + 2], acc[j][g4 * 4 + 3]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_raw, (int)(pok ? (pix * (unsigned)p.raw_ld + (unsigned)ch) * 4u : OOB), 0, 0);
                float4 t = v;
                if (p.scale) {
                    t.x = v.x * sc[g4].x + sh[g4].x; t.y = v.y * sc[g4].y + sh[g4].y; t.z = v.z * sc[g4].z + sh[g4].z; t.w = v.w * sc[g4].w + sh[g4].w;
                }
                if (p.act == CP_ACT_RELU) {
                    t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f);
                } else if (p.act == CP_ACT_LEAKY01) {
                    t.x = fmaxf(t.x, 0.f) - fmaxf(-0.1f * t.x, 0.f); t.y = fmaxf(t.y, 0.f) - fmaxf(-0.1f * t.y, 0.f);
                    t.z = fmaxf(t.z, 0.f) - fmaxf(-0.1f * t.z, 0.f); t.w = fmaxf(t.w, 0.f) - fmaxf(-0.1f * t.w, 0.f);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), r_act, (int)(pok ? (pix * (unsigned)p.act_ld + (unsigned)ch) * 4u : OOB), 0, 0);
            }
        }
        CP_BARRIER();  // tile done: its halo stage may be refilled
    }
}

}  // namespace

namespace cp {

bool stem_applicable(const cp_conv_desc* d) {
    if (!d->weights_halo || d->kh != 7 || d->kw != 7 || d->stride != 2 || d->dilation != 1 || d->pad != 3 || d->cout != 64) return false;
    if (d->num_sources != 1 || d->src[0].channels != 4 || d->src[0].ld != 4 || d->src[0].mode != CP_SRC_DIRECT) return false;
    if (d->tap_label || d->row_scale || d->residual || d->epi_label || d->head_out || d->group_rows) return false;
    if ((d->out_raw && d->out_raw_ld % 4) || (d->out_act && d->out_act_ld % 4)) return false;
    if ((((uintptr_t)d->out_raw) | ((uintptr_t)d->out_act)) & 15) return false;
    return true;
}

int launch_stem_conv(const cp_conv_desc* d, hipStream_t st) {
    StemK k{};
    k.img = d->src[0].data;
    const long long nbytes = (long long)d->batch * d->in_h * d->in_w * 16;
    CP_REQUIRE(nbytes < (1LL << 31), "cp_conv2d_fwd_f32(stem): the image batch spans >= 2 GiB");
    k.img_bytes = (unsigned)nbytes;
    k.pre_scale = d->src[0].pre_scale; k.pre_shift = d->src[0].pre_shift;
    k.W = d->weights_halo;
    k.scale = d->scale; k.shift = d->shift; k.act = d->act;
    k.out_raw = d->out_raw; k.raw_ld = d->out_raw_ld; k.out_act = d->out_act; k.act_ld = d->out_act_ld;
    k.B = d->batch; k.H = d->in_h; k.Wd = d->in_w; k.Ho = d->out_h; k.Wo = d->out_w;
    k.tiles_y = (k.Ho + TH - 1) / TH; k.tiles_x = (k.Wo + TW - 1) / TW;
    k.ntiles = k.B * k.tiles_y * k.tiles_x;
    const size_t lds = (size_t)2 * HP * 4 * sizeof(float);
    int grid = 512;
    if (grid > k.ntiles) grid = k.ntiles;
    CP_LAUNCH(conv_stem_kernel, dim3(grid), dim3(512), lds, st, k);
    return cp::check_launch("cp_conv2d_fwd_f32(stem)");
}

}  // namespace cp

// [25 steps][2 halves][64 cout][4]: W[ky][kx][c][co] of tap 2*step + half, channel c (the 4th channel and tap 49 are zero)
extern "C" int cp_conv_pack_weights_stem_host(const float* w, int layout, int real_channels, float* dst) {
    CP_REQUIRE(w && dst && real_channels >= 1 && real_channels <= 4 && (layout == 0 || layout == 1), "cp_conv_pack_weights_stem_host: bad arguments");
    for (int i = 0; i < NSTEP * 2 * 64 * 4; ++i) dst[i] = 0.f;
    for (int s = 0; s < NSTEP; ++s)
        for (int half = 0; half < 2; ++half) {
            const int t = 2 * s + half;
            if (t >= 49) continue;
            const int ky = t / 7, kx = t % 7;
            for (int co = 0; co < 64; ++co)
                for (int c = 0; c < real_channels; ++c) {
                    const size_t src = (layout == 0) ? ((((size_t)ky * 7 + kx) * real_channels + c) * 64 + co) : ((((size_t)c * 7 + ky) * 7 + kx) * 64 + co);
                    dst[(((size_t)s * 2 + half) * 64 + co) * 4 + c] = w[src];
                }
        }
    return CP_OK;
}
