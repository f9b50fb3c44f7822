// Projected peripheral-feature tables, forward and backward, one launch each (gfx950).
// Contract: include/kpgnn.h, kpgnn_enc_tables_fwd / _bwd.
//
// The bodies build the peripheral features as  gate * Linear(cat_c Emb_c[i_c])  per encoder (models/GNNs.py:172-179 /
// :393-400 / :637-644, layers/feature_encoder.py:62-67).  Since Linear(cat_c e_c) = sum_c e_c W_c^T + b, the features are a
// gather-sum over PROJECTED tables  T_c = gate * Emb_c.weight @ W_c^T  (W_c = the c-th [H, H] column block of
// proj.weight), which depend on the parameters only - 413 rows x H for the reference's nine components.  Computing
// them with framework ops took ~14 launches forward and ~28 backward of ~4.7 us each (tiny tensors: 0.2 ms of every
// step at ANY batch size); here each direction is one launch:
//   fwd  block r < R: table[r,:] = gate_e * E[r,:] @ W_c^T;   block R: bias[:] = sum_e gate_e * mult_e * b_e[:]
//   bwd  block r < R: dE[r,:] = gate_e * gtable[r,:] @ W_c;   blocks (c, o): dW[o, cH:(c+1)H] = gate_e * sum_{r in c}
//        gtable[r,o] * E[r,:];   block per encoder: d(gate_raw) and d(b_e)  (the squashing's derivative included).
// All sums run in a fixed order (bitwise reproducible).
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

constexpr int kMaxComp = 16, kMaxEnc = 4;

struct EncComp { const float* emb; float* gemb; int rows, row0, enc, col_block; };
struct EncEnc { const float* w; const float* b; const float* gate_raw; float* gw; float* gb; float* ggate; int ncomp; float mult; int squash; int row0, rows; };
struct EncParams {
    int H, ncomp, nenc, R;
    EncComp comp[kMaxComp];
    EncEnc enc[kMaxEnc];
    float* table; float* pre; float* bias;     // fwd outputs (pre = the tables before the gate, kept for the backward)
    const float* gtable; const float* gbias;   // bwd inputs (and pre)
};

__device__ __forceinline__ float squash(float v, int kind) { return kind == 1 ? tanhf(v) : 1.0f / (1.0f + __expf(-v)); }
__device__ __forceinline__ float squash_grad(float v, int kind) {
    if (kind == 1) { const float t = tanhf(v); return 1.0f - t * t; }
    const float s = 1.0f / (1.0f + __expf(-v));
    return s * (1.0f - s);
}

__device__ __forceinline__ int comp_of_row(const EncParams& p, int r) {
    int c = 0;
    for (int i = 1; i < p.ncomp; ++i) if (r >= p.comp[i].row0) c = i;
    return c;
}

__global__ void __launch_bounds__(256) enc_tables_fwd_kernel(const EncParams p) {
    __shared__ float erow[256];
    const int H = p.H, r = blockIdx.x, o = threadIdx.x;
    if (r < p.R) {
        const int c = comp_of_row(p, r);
        const EncComp& cc = p.comp[c];
        const EncEnc& e = p.enc[cc.enc];
        if (o < H) erow[o] = cc.emb[(int64_t)(r - cc.row0) * H + o];
        __syncthreads();
        if (o < H) {
            const float g = squash(e.gate_raw[0], e.squash);
            const float* w = e.w + (int64_t)o * e.ncomp * H + (int64_t)cc.col_block * H;   // W[o, cH : (c+1)H]
            float s = 0.f;
            for (int h = 0; h < H; ++h) s = fmaf(erow[h], w[h], s);
            p.pre[(int64_t)r * H + o] = s;
            p.table[(int64_t)r * H + o] = g * s;
        }
    } else if (o < H) {
        float s = 0.f;
        for (int q = 0; q < p.nenc; ++q) s = fmaf(squash(p.enc[q].gate_raw[0], p.enc[q].squash) * p.enc[q].mult, p.enc[q].b[o], s);
        p.bias[o] = s;
    }
}

// grid: R row blocks, then sum_e ncomp_e * H weight blocks, then nenc encoder blocks
__global__ void __launch_bounds__(256) enc_tables_bwd_kernel(const EncParams p) {
    __shared__ float buf[256];
    __shared__ float red[256];
    const int H = p.H, t = threadIdx.x;
    int b = blockIdx.x;
    if (b < p.R) {                                    // dE[r,h] = g * sum_o gtable[r,o] * W[o, cH + h]
        const int r = b, c = comp_of_row(p, r);
        const EncComp& cc = p.comp[c];
        const EncEnc& e = p.enc[cc.enc];
        if (t < H) buf[t] = p.gtable[(int64_t)r * H + t];
        __syncthreads();
        if (t < H) {
            const float g = squash(e.gate_raw[0], e.squash);
            const float* w = e.w + (int64_t)cc.col_block * H + t;
            float s = 0.f;
            for (int o = 0; o < H; ++o) s = fmaf(buf[o], w[(int64_t)o * e.ncomp * H], s);
            cc.gemb[(int64_t)(r - cc.row0) * H + t] = g * s;
        }
        return;
    }
    b -= p.R;
    if (b < p.ncomp * H) {                            // dW[o, cH + h] = g * sum_{r in comp c} gtable[r,o] * E[r,h]
        const int c = b / H, o = b - c * H;
        const EncComp& cc = p.comp[c];
        const EncEnc& e = p.enc[cc.enc];
        if (t < H) {
            const float g = squash(e.gate_raw[0], e.squash);
            float s = 0.f;
            for (int r = 0; r < cc.rows; ++r) s = fmaf(p.gtable[(int64_t)(cc.row0 + r) * H + o], cc.emb[(int64_t)r * H + t], s);
            e.gw[(int64_t)o * e.ncomp * H + (int64_t)cc.col_block * H + t] = g * s;
        }
        return;
    }
    b -= p.ncomp * H;
    if (b < p.nenc) {                                 // gate and bias gradients of encoder b
        const EncEnc& e = p.enc[b];
        const float raw = e.gate_raw[0];
        const float g = squash(raw, e.squash);
        // d/dg = sum_{r in e, o} gtable[r,o] * pre[r,o] + sum_o gbias[o] * mult * b[o]   (pre: the un-gated tables of the forward)
        // (all 256 threads stride over the encoder's rows * H elements, eight independent loads per trip: a per-column loop
        //  over the rows was one dependent load chain of ~360 steps - 90 us for a 150 KB reduction)
        float acc = 0.f;
        {
            const int64_t base = (int64_t)e.row0 * H, n = (int64_t)e.rows * H;
            float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int64_t i = t;
            for (; i + 7 * 256 < n; i += 8 * 256) {
#pragma unroll
                for (int u = 0; u < 8; ++u) a8[u] = fmaf(p.gtable[base + i + u * 256], p.pre[base + i + u * 256], a8[u]);
            }
            for (; i < n; i += 256) a8[0] = fmaf(p.gtable[base + i], p.pre[base + i], a8[0]);
            acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
        }
        if (t < H) {
            acc = fmaf(p.gbias[t] * e.mult, e.b[t], acc);
            e.gb[t] = g * e.mult * p.gbias[t];
        }
        red[t] = acc;
        __syncthreads();
        if (t == 0) {
            float tot = 0.f;
            for (int i = 0; i < 256; ++i) tot += red[i];
            e.ggate[0] = tot * squash_grad(raw, e.squash);
        }
    }
}

int fill(const kpgnn_enc_tables_desc* d, EncParams* p, bool bwd) {
    KPGNN_REQUIRE(d != nullptr, "enc_tables: NULL descriptor");
    KPGNN_REQUIRE(d->H >= 1 && d->H <= 256 && d->num_components >= 1 && d->num_components <= kMaxComp &&
                  d->num_encoders >= 1 && d->num_encoders <= kMaxEnc, "enc_tables: bad H=%d components=%d encoders=%d",
                  d->H, d->num_components, d->num_encoders);
    p->H = d->H; p->ncomp = d->num_components; p->nenc = d->num_encoders;
    int row = 0;
    int ncomp_of[kMaxEnc] = {0, 0, 0, 0};
    for (int c = 0; c < d->num_components; ++c) {
        const int e = d->comp_encoder[c];
        KPGNN_REQUIRE(e >= 0 && e < d->num_encoders && d->comp_emb[c] && d->comp_rows[c] >= 1, "enc_tables: bad component %d", c);
        KPGNN_REQUIRE(!bwd || d->comp_gemb[c], "enc_tables_bwd: NULL gradient buffer of component %d", c);
        p->comp[c].emb = d->comp_emb[c]; p->comp[c].gemb = d->comp_gemb[c]; p->comp[c].rows = d->comp_rows[c];
        p->comp[c].row0 = row; p->comp[c].enc = e; p->comp[c].col_block = ncomp_of[e]++;
        row += d->comp_rows[c];
    }
    p->R = row;
    for (int e = 0; e < d->num_encoders; ++e) {     // (the components of an encoder are consecutive: its rows are one range)
        int r0 = -1, rows = 0;
        for (int c = 0; c < d->num_components; ++c)
            if (p->comp[c].enc == e) {
                if (r0 < 0) r0 = p->comp[c].row0;
                KPGNN_REQUIRE(p->comp[c].row0 == r0 + rows, "enc_tables: the components of encoder %d are not consecutive", e);
                rows += p->comp[c].rows;
            }
        p->enc[e].row0 = r0 < 0 ? 0 : r0; p->enc[e].rows = rows;
        KPGNN_REQUIRE(d->enc_w[e] && d->enc_b[e] && d->enc_gate[e] && ncomp_of[e] >= 1, "enc_tables: bad encoder %d", e);
        KPGNN_REQUIRE(!bwd || (d->enc_gw[e] && d->enc_gb[e] && d->enc_ggate[e]), "enc_tables_bwd: NULL gradient buffer of encoder %d", e);
        p->enc[e].w = d->enc_w[e]; p->enc[e].b = d->enc_b[e]; p->enc[e].gate_raw = d->enc_gate[e];
        p->enc[e].gw = d->enc_gw[e]; p->enc[e].gb = d->enc_gb[e]; p->enc[e].ggate = d->enc_ggate[e];
        p->enc[e].ncomp = ncomp_of[e]; p->enc[e].mult = d->enc_mult[e]; p->enc[e].squash = d->enc_squash[e];
    }
    p->table = d->table; p->pre = d->pre; p->bias = d->bias; p->gtable = d->gtable; p->gbias = d->gbias;
    return KPGNN_OK;
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" int kpgnn_enc_tables_fwd(const kpgnn_enc_tables_desc* d, kpgnn_stream_t stream) {
    EncParams p;
    int rc = fill(d, &p, false);
    if (rc != KPGNN_OK) return rc;
    KPGNN_REQUIRE(d->table && d->pre && d->bias, "enc_tables_fwd: NULL table/pre/bias");
    hipLaunchKernelGGL(enc_tables_fwd_kernel, dim3(p.R + 1), dim3(256), 0, (hipStream_t)stream, p);
    KPGNN_LAUNCH_CHECK("enc_tables_fwd_kernel");
    return KPGNN_OK;
}

extern "C" int kpgnn_enc_tables_bwd(const kpgnn_enc_tables_desc* d, kpgnn_stream_t stream) {
    EncParams p;
    int rc = fill(d, &p, true);
    if (rc != KPGNN_OK) return rc;
    KPGNN_REQUIRE(d->gtable && d->gbias && d->pre, "enc_tables_bwd: NULL gtable/gbias/pre");
    hipLaunchKernelGGL(enc_tables_bwd_kernel, dim3(p.R + p.ncomp * p.H + p.nenc), dim3(256), 0, (hipStream_t)stream, p);
    KPGNN_LAUNCH_CHECK("enc_tables_bwd_kernel");
    return KPGNN_OK;
}
