// Gradient of the peripheral dictionary under the fused geometric combine (gfx950).  Contract: include/kpgnn.h,
// kpgnn_dict_grad.
//
//   gdict[u,:] = sum_k theta[k,:] * M[u,k,:],     M[u,k,:] = sum over nodes i with uid[i,k] == u of gh[i,:]
//
// (the dictionary row P[uid[i,k]] enters h_i = sum_k theta_k * (act(S_ik) + P_ik) linearly: KPGINplus.py:74-88 with the
// peripheral features of models/GNNs.py:393-400 in their dictionary form).  Only gh [N,D] and the ids are read - not the
// [N,K,D] gradient the edge-code tables need - so this is a 20 MB stream at the bench shape, against 158 MB when the
// same sums ride along in kpgnn_table_grad's walk (where they cost 25-56 us of a launch).
//
// Wave w of a block owns hop w: it walks the block's contiguous run of nodes in order, keeps the sum of a run of equal
// ids in registers (lane = two feature columns) and adds it to ITS accumulator row (u, w) in LDS with a plain
// read-modify-write when the id changes.  No row is shared between waves, every sum has one fixed order (bitwise
// reproducible), no atomics.  gh is staged through LDS once per block (all hops read the same rows) with the next
// chunk's loads in flight.  At the end the block multiplies by theta, adds the hops in order and leaves [U,D] in its
// slab row; a second launch adds the slabs in block order.
//
// With `dom` (one designated id per hop): in a molecule batch ONE id covers 82-99.9 % of the nodes of a hop.  Its sum is
// not accumulated but obtained as  (sum of ALL gh rows of the block) - (sum of the hop's other ids): the total is one
// row-add per node shared by all hops (the waves split the rows), and a wave only walks the nodes whose id at its hop is NOT
// the designated one (a ballot over the chunk's 64 ids) - ~2 row-adds per node instead of 8, and no scalar chain over
// the dominant nodes (and no LDS staging: the few rows a wave needs come straight from L2).  Exact for any choice of
// `dom`; fixed orders everywhere.
#include "kpgnn_common.h"

namespace kpgnn {
namespace {

constexpr int kWavesDG = 8;
constexpr int kThreadsDG = kWave * kWavesDG;
constexpr int kChunk = 64;                  // nodes per staged chunk of gh

struct DgParams {
    const int32_t* n_dyn;
    int N, K, D, U;
    const int32_t* uid; int64_t uid_stride;
    const float* theta;
    const float* gh;
    float* slab;                            // [gridDim.x][U][D]
    const int32_t* dom;                     // optional [K]: a designated id per hop (any id is correct; the most frequent one pays)
};

// LDS (floats): acc [U*K][D] | ghs [2][kChunk][D]
__global__ void __launch_bounds__(kThreadsDG)
dict_grad_kernel(DgParams p) {
    p.N = live_rows(p.N, p.n_dyn);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int lane = threadIdx.x & (kWave - 1);
    const int D = p.D, K = p.K, U = p.U;
    const int c = lane * 2;
    const bool col_ok = c < D;                        // D is even
    const int cc = col_ok ? c : 0;
    float* acc = lds;
    float* ghs = lds + U * K * D;
    // (16-byte stores: with one float per store this fill was 40 % of the launch - 6,000 of 15,000 cycles, s_memtime per phase)
    {
        const int n4 = (U * K * D) & ~3;
        for (int i = threadIdx.x * 4; i < n4; i += kThreadsDG * 4) *reinterpret_cast<float4*>(acc + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = n4 + threadIdx.x; i < U * K * D; i += kThreadsDG) acc[i] = 0.f;
    }
    const int per = (p.N + gridDim.x - 1) / gridDim.x;
    const int n0 = blockIdx.x * per, n1 = min(p.N, n0 + per);
    const int chunk_floats = kChunk * D;              // multiple of 4 (D even, kChunk 64)
    // chunk staging: thread t copies float4 t, t + 512, ... of the chunk (kChunk*D/4 <= 2048 float4: D <= 128)
    constexpr int kPre = 4;
    float4 pre[kPre];
    auto load_chunk = [&](int node0) {
#pragma unroll
        for (int q = 0; q < kPre; ++q) {
            const int i = (q * kThreadsDG + threadIdx.x) * 4;
            pre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < chunk_floats && node0 + i / D < n1)     // (D % 4 == 0 on this path: a float4 never straddles two rows)
                pre[q] = *reinterpret_cast<const float4*>(p.gh + (int64_t)node0 * D + i);
        }
    };
    auto load_uid = [&](int node0) -> int {          // lane l < kChunk: id of node node0 + l at this wave's hop
        int v = -1;
        if (w < K && lane < kChunk && node0 + lane < n1) v = p.uid[(int64_t)(node0 + lane) * p.uid_stride + w];
        return v;
    };
    const bool vec_ok = (D % 4 == 0) && (((uintptr_t)p.gh & 15) == 0);
    int cur = -1;
    float ra = 0.f, rb = 0.f;                         // the running sum of the current id (two columns)
    bool pending = false;                             // a finished run whose accumulator row is being read
    float* pq = acc;
    float2 pold = make_float2(0.f, 0.f), psum = make_float2(0.f, 0.f);
    auto settle = [&]() {
        if (pending) { if (col_ok) *reinterpret_cast<float2*>(pq) = make_float2(pold.x + psum.x, pold.y + psum.y); pending = false; }
    };
    auto leave = [&]() {
        settle();
        pq = acc + (cur * K + w) * D + cc;
        pold = *reinterpret_cast<const float2*>(pq);
        psum = make_float2(ra, rb);
        pending = true;
    };
    int uidv = 0, nuid = 0;
    const bool use_dom = p.dom != nullptr;
    if (n0 < n1) {
        if (vec_ok && !use_dom) load_chunk(n0);
        nuid = load_uid(n0);
    }
    int domk = -2;                                    // (never equals an id)
    if (use_dom && w < K) { domk = p.dom[w]; if (domk < 0 || domk >= U) domk = 0; }
    float ta = 0.f, tb = 0.f;                         // this wave's share of the block's total (rows j = w mod 8 of every chunk)
    if (use_dom) {
        // No staging and no barriers: a wave reads the few rows it needs (its eighth of the chunk for the total, the nodes whose
        // id at its hop is not the designated one) straight from L2.  A block's ~185 nodes are ~3 chunks: the ids and the total's
        // rows of up to FOUR chunks are requested before anything is used (unconditional, clamped; one round trip instead of one
        // per chunk: 19 -> 14 us per launch), only the few non-designated rows of a chunk depend on its ids.
        constexpr int NCH = 4, RPW = kChunk / kWavesDG;
        const int nlast = n1 - 1;                      // (n0 < n1 whenever anything is loaded)
        for (int base = n0; base < n1; base += NCH * kChunk) {
            int uids[NCH];
            float2 tvs[NCH][RPW];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
                uids[ch] = p.uid[(int64_t)min(base + ch * kChunk + lane, nlast) * p.uid_stride + (w < K ? w : 0)];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int q = 0; q < RPW; ++q)
                    tvs[ch][q] = *reinterpret_cast<const float2*>(p.gh + (int64_t)min(base + ch * kChunk + q * kWavesDG + w, nlast) * D + cc);
            if (base == n0) __syncthreads();           // the accumulator rows are zero (their stores went out under the requests)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int node0 = base + ch * kChunk;
                if (node0 >= n1) break;                // (uniform)
                const int nn = min(kChunk, n1 - node0);
                uidv = (w < K && lane < nn) ? uids[ch] : -1;
                const float* gb = p.gh + (int64_t)node0 * D + cc;
                unsigned long long todo = (w < K) ? __ballot(uidv >= 0 && uidv != domk) : 0ull;
                while (todo) {
                    int js[8]; float2 v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        js[q] = todo ? (int)__builtin_ctzll(todo) : -1;
                        if (todo) todo &= todo - 1;
                        v[q] = js[q] >= 0 ? *reinterpret_cast<const float2*>(gb + (int64_t)js[q] * D) : make_float2(0.f, 0.f);
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        if (js[q] >= 0) {
                            const int u = __builtin_amdgcn_readlane(uidv, js[q]);
                            if (u != cur) {                        // wave-uniform
                                if (cur >= 0) leave();
                                cur = u;
                                ra = rb = 0.f;
                            }
                            ra += v[q].x; rb += v[q].y;
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < RPW; ++q)
                    if (q * kWavesDG + w < nn) { ta += tvs[ch][q].x; tb += tvs[ch][q].y; }
            }
        }
        if (n0 >= n1) __syncthreads();                 // (a block without nodes still meets the barrier count)
    }
    int buf = 0;
    for (int node0 = use_dom ? n1 : n0; node0 < n1; node0 += kChunk, buf ^= 1) {
        float* gs = ghs + buf * chunk_floats;
        uidv = nuid;
        if (vec_ok) {
#pragma unroll
            for (int q = 0; q < kPre; ++q) {
                const int i = (q * kThreadsDG + threadIdx.x) * 4;
                if (i < chunk_floats) *reinterpret_cast<float4*>(gs + i) = pre[q];
            }
            if (node0 + kChunk < n1) load_chunk(node0 + kChunk);
        } else {
            const int nn = min(kChunk, n1 - node0);
            for (int i = threadIdx.x; i < nn * D; i += kThreadsDG) gs[i] = p.gh[(int64_t)node0 * D + i];
        }
        nuid = load_uid(node0 + kChunk);
        __syncthreads();       // chunk staged; (two buffers: the previous chunk's readers are at most one barrier behind)
        if (w < K) {
#pragma unroll 1
            for (int j0 = 0; j0 < kChunk; j0 += 8) {
                int u[8]; float2 v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    u[q] = __builtin_amdgcn_readlane(uidv, j0 + q);
                    v[q] = *reinterpret_cast<const float2*>(gs + (j0 + q) * D + cc);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (u[q] >= 0) {
                        if (u[q] != cur) {                         // wave-uniform
                            if (cur >= 0) leave();
                            cur = u[q];
                            ra = rb = 0.f;
                        }
                        ra += v[q].x; rb += v[q].y;
                    }
                }
            }
        }
    }
    if (cur >= 0) leave();
    settle();
    __syncthreads();
    if (use_dom) {
        // the designated rows: acc[dom_k, k, :] = (total of the block) - (the hop's other ids), waves and ids in order
        float* tot = ghs;                             // [kWavesDG][D], then [D]   (the staging buffers are free now)
        if (col_ok) *reinterpret_cast<float2*>(tot + w * D + cc) = make_float2(ta, tb);
        __syncthreads();
        for (int d = threadIdx.x; d < D; d += kThreadsDG) {
            float t = 0.f;
            for (int q = 0; q < kWavesDG; ++q) t += tot[q * D + d];
            tot[kWavesDG * D + d] = t;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < K * D; i += kThreadsDG) {
            const int k = i / D, d = i - k * D;
            int dk = p.dom[k];
            if (dk < 0 || dk >= U) dk = 0;
            // (four interleaved partial sums, ids u = j mod 4 each, added in a fixed order: the reads of one chain of U dependent
            //  adds were 4,100 cycles)
            float r4[4] = {0.f, 0.f, 0.f, 0.f};
            const float* col = acc + k * D + d;
            int u = 0;
            for (; u + 3 < U; u += 4) {
                const float a0 = col[(u + 0) * K * D], a1 = col[(u + 1) * K * D], a2 = col[(u + 2) * K * D], a3 = col[(u + 3) * K * D];
                r4[0] += (u + 0 != dk) ? a0 : 0.f; r4[1] += (u + 1 != dk) ? a1 : 0.f;
                r4[2] += (u + 2 != dk) ? a2 : 0.f; r4[3] += (u + 3 != dk) ? a3 : 0.f;
            }
            for (; u < U; ++u) r4[u & 3] += (u != dk) ? col[u * K * D] : 0.f;
            acc[(dk * K + k) * D + d] = tot[kWavesDG * D + d] - ((r4[0] + r4[1]) + (r4[2] + r4[3]));
        }
        __syncthreads();
    }
    // gdict_block[u, d] = sum_k theta[k, d] * acc[u, k, d], hops in order.  theta goes to LDS once (a global read per term made
    // every term of the K-chain a round trip: 4,200 cycles), all K terms of an output are read before the chain starts.
    float* th = ghs;                                  // [K][D]  (the staging buffers are free; K D <= 2 kChunk D)
    __syncthreads();
    for (int i = threadIdx.x; i < K * D; i += kThreadsDG) th[i] = p.theta[i];
    __syncthreads();
    for (int i = threadIdx.x; i < U * D; i += kThreadsDG) {
        const int u = i / D, d = i - u * D;
        float a[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = k < K ? acc[(u * K + k) * D + d] * th[k * D + d] : 0.f;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) if (k < K) s += a[k];
        p.slab[(int64_t)blockIdx.x * U * D + i] = s;
    }
}

// ---- the same gradient for ALL the layers that read one dictionary, in one launch:
//   gdict[u,:] = sum_l sum_{k < K_l} theta_l[k,:] * sum over nodes i with uid[i,k] == u of gh_l[i,:]
// A launch of the kernel above is mostly its fixed passes - filling the [U K][D] accumulator table, the designated-row fix, the
// final hop sum: ~14,000 of its 18,000 cycles, one block per CU - and a sequential stack runs it once per layer over the SAME
// ids.  Here the table is filled once, every layer's rows are added into it already multiplied by that layer's theta (so the
// final pass is a plain sum over hops), and the fix runs once with  T_k = sum_l theta_l[k] * (block total of gh_l).
// Designated ids are required (the variant that reads the few non-designated rows straight from L2).
constexpr int kMaxDgLayers = 16;
struct DgMultiParams {
    const int32_t* n_dyn;
    int N, K, D, U, L;                      // K = the largest K_l
    const int32_t* uid; int64_t uid_stride;
    const float* theta[kMaxDgLayers]; const float* gh[kMaxDgLayers]; int Kl[kMaxDgLayers];
    float* slab;                            // [gridDim.x][U][D]
    const int32_t* dom;                     // [K]
};

// LDS (floats): acc [U*K][D] | tot [L][kWavesDG][D] | totl [L][D]
__global__ void __launch_bounds__(kThreadsDG)
dict_grad_multi_kernel(DgMultiParams p) {
    p.N = live_rows(p.N, p.n_dyn);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int lane = threadIdx.x & (kWave - 1);
    const int D = p.D, K = p.K, U = p.U, L = p.L;
    const int c = lane * 2;
    const bool col_ok = c < D;                        // D is even
    const int cc = col_ok ? c : 0;
    float* acc = lds;
    float* tot = lds + U * K * D;
    float* totl = tot + L * kWavesDG * D;
    {
        const int n4 = (U * K * D) & ~3;
        for (int i = threadIdx.x * 4; i < n4; i += kThreadsDG * 4) *reinterpret_cast<float4*>(acc + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = n4 + threadIdx.x; i < U * K * D; i += kThreadsDG) acc[i] = 0.f;
    }
    const int per = (p.N + gridDim.x - 1) / gridDim.x;
    const int n0 = blockIdx.x * per, n1 = min(p.N, n0 + per);
    int cur = -1;
    float ra = 0.f, rb = 0.f;
    bool pending = false;
    float* pq = acc;
    float2 pold = make_float2(0.f, 0.f), psum = make_float2(0.f, 0.f);
    auto settle = [&]() {
        if (pending) { if (col_ok) *reinterpret_cast<float2*>(pq) = make_float2(pold.x + psum.x, pold.y + psum.y); pending = false; }
    };
    auto leave = [&]() {
        settle();
        pq = acc + (cur * K + w) * D + cc;
        pold = *reinterpret_cast<const float2*>(pq);
        psum = make_float2(ra, rb);
        pending = true;
    };
    int domk = -2;
    if (w < K) { domk = p.dom[w]; if (domk < 0 || domk >= U) domk = 0; }
    // (Measured alternatives, both slower than this plain loop's 96 us for 8 layers: the first 8 non-designated rows of EVERY chunk
    //  requested up front with the total's rows, unconditionally - 112 us, the clamped requests of the hops that have none cost
    //  more than the trips they save; two chunks at a time with the next layer's requests in flight - 115 us.)
    constexpr int NCH = 4, RPW = kChunk / kWavesDG;
    const int nlast = n1 - 1;
    bool zeroed = false;
    for (int base = n0; base < n1; base += NCH * kChunk) {
        int uids[NCH];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
            uids[ch] = p.uid[(int64_t)min(base + ch * kChunk + lane, nlast) * p.uid_stride + (w < K ? w : 0)];
        for (int l = 0; l < L; ++l) {
            // (the layer's pointers are picked with uniform selects over constant indices: indexing the argument struct with a
            //  runtime value copies it to scratch, and every use becomes a scratch load - 134 us instead of 96)
            const float* gh = p.gh[0]; const float* thp = p.theta[0]; int kl = p.Kl[0];
#pragma unroll
            for (int q = 1; q < kMaxDgLayers; ++q) if (q == l) { gh = p.gh[q]; thp = p.theta[q]; kl = p.Kl[q]; }
            const bool hop_on = w < kl;
            float2 tvs[NCH][RPW];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
                for (int q = 0; q < RPW; ++q)
                    tvs[ch][q] = *reinterpret_cast<const float2*>(gh + (int64_t)min(base + ch * kChunk + q * kWavesDG + w, nlast) * D + cc);
            const float2 th = hop_on ? *reinterpret_cast<const float2*>(thp + w * D + cc) : make_float2(0.f, 0.f);
            if (!zeroed) { __syncthreads(); zeroed = true; }   // the accumulator rows are zero (their stores went out under the requests)
            float ta = 0.f, tb = 0.f;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int node0 = base + ch * kChunk;
                if (node0 >= n1) break;                // (uniform)
                const int nn = min(kChunk, n1 - node0);
                const int uidv = (hop_on && lane < nn) ? uids[ch] : -1;
                const float* gb = gh + (int64_t)node0 * D + cc;
                unsigned long long todo = hop_on ? __ballot(uidv >= 0 && uidv != domk) : 0ull;
                while (todo) {
                    int js[8]; float2 v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        js[q] = todo ? (int)__builtin_ctzll(todo) : -1;
                        if (todo) todo &= todo - 1;
                        v[q] = js[q] >= 0 ? *reinterpret_cast<const float2*>(gb + (int64_t)js[q] * D) : make_float2(0.f, 0.f);
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        if (js[q] >= 0) {
                            const int u = __builtin_amdgcn_readlane(uidv, js[q]);
                            if (u != cur) {                        // wave-uniform
                                if (cur >= 0) leave();
                                cur = u;
                                ra = rb = 0.f;
                            }
                            ra = fmaf(th.x, v[q].x, ra); rb = fmaf(th.y, v[q].y, rb);
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < RPW; ++q)
                    if (q * kWavesDG + w < nn) { ta += tvs[ch][q].x; tb += tvs[ch][q].y; }
            }
            // this wave's share of the block's total of layer l (rows j = w mod 8 of every chunk); one writer per slot
            float* ts = tot + (l * kWavesDG + w) * D + cc;
            if (col_ok) {
                if (base == n0) *reinterpret_cast<float2*>(ts) = make_float2(ta, tb);
                else { const float2 o = *reinterpret_cast<const float2*>(ts); *reinterpret_cast<float2*>(ts) = make_float2(o.x + ta, o.y + tb); }
            }
        }
    }
    if (!zeroed) {                                     // a block without nodes: zero totals, and it still meets the barrier count
        for (int i = threadIdx.x; i < L * kWavesDG * D; i += kThreadsDG) tot[i] = 0.f;
        __syncthreads();
    }
    if (cur >= 0) leave();
    settle();
    __syncthreads();
    for (int i = threadIdx.x; i < L * D; i += kThreadsDG) {
        const int l = i / D, d = i - l * D;
        float t = 0.f;
        for (int q = 0; q < kWavesDG; ++q) t += tot[(l * kWavesDG + q) * D + d];
        totl[i] = t;
    }
    __syncthreads();
    // the designated rows: acc[dom_k, k, :] = sum_l theta_l[k] * (block total of layer l) - (the hop's other ids)
    for (int i = threadIdx.x; i < K * D; i += kThreadsDG) {
        const int k = i / D, d = i - k * D;
        int dk = p.dom[k];
        if (dk < 0 || dk >= U) dk = 0;
        float T = 0.f;
#pragma unroll
        for (int l = 0; l < kMaxDgLayers; ++l)
            if (l < L) T += k < p.Kl[l] ? p.theta[l][k * D + d] * totl[l * D + d] : 0.f;
        float r4[4] = {0.f, 0.f, 0.f, 0.f};
        const float* col = acc + k * D + d;
        int u = 0;
        for (; u + 3 < U; u += 4) {
            const float a0 = col[(u + 0) * K * D], a1 = col[(u + 1) * K * D], a2 = col[(u + 2) * K * D], a3 = col[(u + 3) * K * D];
            r4[0] += (u + 0 != dk) ? a0 : 0.f; r4[1] += (u + 1 != dk) ? a1 : 0.f;
            r4[2] += (u + 2 != dk) ? a2 : 0.f; r4[3] += (u + 3 != dk) ? a3 : 0.f;
        }
        for (; u < U; ++u) r4[u & 3] += (u != dk) ? col[u * K * D] : 0.f;
        acc[(dk * K + k) * D + d] = T - ((r4[0] + r4[1]) + (r4[2] + r4[3]));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < U * D; i += kThreadsDG) {
        const int u = i / D, d = i - u * D;
        float a[kWavesDG];
#pragma unroll
        for (int k = 0; k < kWavesDG; ++k) a[k] = k < K ? acc[(u * K + k) * D + d] : 0.f;
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < kWavesDG; ++k) if (k < K) s2 += a[k];
        p.slab[(int64_t)blockIdx.x * U * D + i] = s2;
    }
}

struct DgPlan { int grid; size_t lds, ws_bytes; };

// has_dom: the launch carries a designated id per hop - that path reads gh straight from L2 and needs no staging buffers
// (only [kWavesDG + 1][D] floats for the block total), so larger dictionaries fit: U = 35 (a 10,000-molecule DATASET's distinct
// tuples, dataset.py) at K = 8, D = 104 takes 120 KB instead of 170.
bool dg_plan(int N, int K, int D, int U, DgPlan* pl, bool has_dom = false) {
    if (N < 1 || K < 1 || K > kWavesDG || D < 2 || D > 2 * kWave || (D & 1) || U < 1) return false;
    pl->lds = sizeof(float) * ((size_t)U * K * D + (has_dom ? (size_t)(kWavesDG + 1) * D : 2 * (size_t)kChunk * D));
    if (pl->lds > 160 * 1024) return false;
    int grid = device_facts().cu_count;
    const int chunks = (N + kChunk - 1) / kChunk;
    if (grid > chunks) grid = chunks;
    pl->grid = grid < 1 ? 1 : grid;
    pl->ws_bytes = sizeof(float) * (size_t)pl->grid * U * D;
    return true;
}

}  // namespace
}  // namespace kpgnn

using namespace kpgnn;

extern "C" size_t kpgnn_dict_grad_workspace_bytes(int32_t N, int32_t K, int32_t D, int32_t n_dict) {
    DgPlan pl;
    // (the workspace does not depend on the variant; a launch WITHOUT a designated id per hop whose staging buffers do not fit
    //  is refused by kpgnn_dict_grad itself: KPGNN_ELIMIT)
    return dg_plan(N, K, D, n_dict, &pl, true) ? pl.ws_bytes : 0;
}

extern "C" int32_t kpgnn_dict_grad_slabs(int32_t N) {
    DgPlan pl;
    return (N >= 1 && dg_plan(N, 1, 2, 1, &pl)) ? pl.grid : 0;
}

extern "C" int kpgnn_dict_grad(const kpgnn_dict_grad_desc* d, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr, "dict_grad: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 0 && d->K >= 1 && d->D >= 1 && d->n_dict >= 1, "dict_grad: bad N=%d K=%d D=%d n_dict=%d", d->N, d->K,
                  d->D, d->n_dict);
    KPGNN_REQUIRE(d->gdict != nullptr || d->defer_reduce, "dict_grad: NULL gdict");
    hipStream_t s = (hipStream_t)stream;
    if (d->N == 0) { if (d->defer_reduce) return fail(KPGNN_EINVAL, "dict_grad: defer_reduce with N == 0"); KPGNN_HIP_TRY(hipMemsetAsync(d->gdict, 0, sizeof(float) * (size_t)d->n_dict * d->D, s)); return KPGNN_OK; }
    DgPlan pl;
    if (!dg_plan(d->N, d->K, d->D, d->n_dict, &pl, d->dominant != nullptr))
        return fail(KPGNN_ELIMIT, "dict_grad: K=%d (<= 8), even D=%d (<= 128) and n_dict*K*D*4 + staging <= 160 KB of LDS needed "
                    "(n_dict=%d); use kpgnn_table_grad's dictionary path", d->K, d->D, d->n_dict);
    KPGNN_REQUIRE(d->uid && d->theta && d->gh && d->uid_stride >= d->K, "dict_grad: NULL uid/theta/gh or uid_stride < K");
    KPGNN_REQUIRE(d->workspace && d->workspace_bytes >= pl.ws_bytes, "dict_grad: workspace too small (%zu < %zu)",
                  (size_t)d->workspace_bytes, pl.ws_bytes);
    DgParams p;
    p.N = d->N; p.n_dyn = d->n_dyn; p.K = d->K; p.D = d->D; p.U = d->n_dict;
    p.uid = d->uid; p.uid_stride = d->uid_stride; p.theta = d->theta; p.gh = d->gh; p.slab = (float*)d->workspace;
    p.dom = d->dominant;
    if (pl.lds > 64 * 1024) KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)dict_grad_kernel, pl.lds));
    hipLaunchKernelGGL(dict_grad_kernel, dim3(pl.grid), dim3(kThreadsDG), pl.lds, s, p);
    KPGNN_LAUNCH_CHECK("dict_grad_kernel");
    if (d->defer_reduce) return KPGNN_OK;
    return slab_reduce(p.slab, pl.grid, (int64_t)p.U * p.D, d->gdict, (int64_t)p.U * p.D, nullptr, 0, nullptr, s);
}

extern "C" int kpgnn_dict_grad_multi(const kpgnn_dict_grad_multi_desc* d, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr, "dict_grad_multi: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 1 && d->D >= 1 && d->n_dict >= 1 && d->L >= 1, "dict_grad_multi: bad N=%d D=%d n_dict=%d L=%d", d->N, d->D,
                  d->n_dict, d->L);
    if (d->L > kMaxDgLayers) return fail(KPGNN_ELIMIT, "dict_grad_multi: L=%d > %d layers", d->L, kMaxDgLayers);
    KPGNN_REQUIRE(d->uid && d->dominant && d->gdict && d->workspace, "dict_grad_multi: NULL uid / dominant / gdict / workspace");
    int K = 0;
    for (int l = 0; l < d->L; ++l) {
        KPGNN_REQUIRE(d->theta[l] && d->gh[l] && d->K[l] >= 1, "dict_grad_multi: layer %d: NULL theta / gh or K < 1", l);
        if (d->K[l] > K) K = d->K[l];
    }
    KPGNN_REQUIRE(d->uid_stride >= K, "dict_grad_multi: uid_stride < K");
    DgPlan pl;
    if (!dg_plan(d->N, K, d->D, d->n_dict, &pl, true))
        return fail(KPGNN_ELIMIT, "dict_grad_multi: K=%d (<= 8), even D=%d (<= 128) needed", K, d->D);
    pl.lds = sizeof(float) * ((size_t)d->n_dict * K * d->D + (size_t)d->L * (kWavesDG + 1) * d->D);
    if (pl.lds > 160 * 1024) return fail(KPGNN_ELIMIT, "dict_grad_multi: %zu bytes of LDS needed (n_dict=%d, L=%d)", pl.lds, d->n_dict, d->L);
    KPGNN_REQUIRE(d->workspace_bytes >= pl.ws_bytes, "dict_grad_multi: workspace too small (%zu < %zu)", (size_t)d->workspace_bytes, pl.ws_bytes);
    DgMultiParams p;
    p.N = d->N; p.n_dyn = d->n_dyn; p.K = K; p.D = d->D; p.U = d->n_dict; p.L = d->L;
    p.uid = d->uid; p.uid_stride = d->uid_stride; p.slab = (float*)d->workspace; p.dom = d->dominant;
    for (int l = 0; l < kMaxDgLayers; ++l) {
        p.theta[l] = l < d->L ? d->theta[l] : nullptr; p.gh[l] = l < d->L ? d->gh[l] : nullptr; p.Kl[l] = l < d->L ? d->K[l] : 0;
    }
    hipStream_t s = (hipStream_t)stream;
    if (pl.lds > 64 * 1024) KPGNN_HIP_TRY(ensure_dynamic_lds((const void*)dict_grad_multi_kernel, pl.lds));
    hipLaunchKernelGGL(dict_grad_multi_kernel, dim3(pl.grid), dim3(kThreadsDG), pl.lds, s, p);
    KPGNN_LAUNCH_CHECK("dict_grad_multi_kernel");
    // gdict = (gdict_acc +) the slabs in block order
    return slab_reduce(p.slab, pl.grid, (int64_t)p.U * p.D, d->gdict, (int64_t)p.U * p.D, nullptr, 0, nullptr, s);
}
