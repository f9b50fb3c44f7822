// C-ABI entry points of the fused Linear + BatchNorm kernels (lin_fused.h) and the capture-id helper.
#include "lin_fused.h"

using namespace kpgnn;

extern "C" int kpgnn_linear_bn(const kpgnn_linear_bn_desc* d, kpgnn_stream_t stream) {
    KPGNN_REQUIRE(d != nullptr, "linear_bn: NULL descriptor");
    KPGNN_REQUIRE(d->N >= 1 && d->O >= 1 && d->I >= 1, "linear_bn: bad N=%lld O=%d I=%d", (long long)d->N, d->O, d->I);
    KPGNN_REQUIRE(d->x && d->w && d->y, "linear_bn: NULL pointer");
    KPGNN_REQUIRE(d->pro >= 0 && d->pro <= 3 && d->epi >= 0 && d->epi <= 2, "linear_bn: bad pro=%d / epi=%d", d->pro, d->epi);
    KPGNN_REQUIRE(d->pro < 2 || d->bias == nullptr, "linear_bn: the backward variants (pro %d) take no bias", d->pro);
    if (!lin_supported_width(d->I) || (d->O % 4) != 0 || d->O > 128)
        return fail(KPGNN_ELIMIT, "linear_bn: I=%d must be one of 32, 64, 96, 104, 128 and O=%d a multiple of 4 <= 128", d->I, d->O);
    uintptr_t al = (uintptr_t)d->x | (uintptr_t)d->y | (uintptr_t)d->bias | (uintptr_t)d->x2 | (uintptr_t)d->xt | (uintptr_t)d->e_x;
    if (!d->w_transposed) al |= (uintptr_t)d->w;
    if (al & 15) return fail(KPGNN_ELIMIT, "linear_bn: operands must be 16-B aligned");
    LinFParams p = {};
    p.N = d->N; p.n_dyn = d->n_dyn; p.O = d->O; p.I = d->I; p.wt = d->w_transposed ? 1 : 0;
    p.x = d->x; p.w = d->w; p.bias = d->bias; p.y = d->y;
    if (d->pro != 0) {
        KPGNN_REQUIRE(d->in_slot && d->in_gamma && d->in_beta && d->in_mean && d->in_invstd, "linear_bn: pro %d needs in_slot, in_gamma, in_beta, in_mean, in_invstd", d->pro);
        p.in_slot = d->in_slot; p.in_gamma = d->in_gamma; p.in_beta = d->in_beta; p.pro_relu = d->pro_relu;
        p.in_eps = d->in_eps; p.momentum = d->momentum; p.in_mean = d->in_mean; p.in_invstd = d->in_invstd;
        p.rmean = d->running_mean; p.rvar = d->running_var; p.nbt = d->num_batches_tracked;
    }
    if (d->pro >= 2) {
        KPGNN_REQUIRE(d->x2 && d->dgamma && d->dbeta, "linear_bn: pro %d needs x2, dgamma, dbeta", d->pro);
        p.x2 = d->x2; p.xt = d->xt; p.dgamma = d->dgamma; p.dbeta = d->dbeta;
    }
    if (d->pro == 3) {
        KPGNN_REQUIRE(d->o_mean && d->o_invstd && d->o_gamma && d->o_dgamma && d->o_dbeta,
                      "linear_bn: pro 3 needs o_mean, o_invstd, o_gamma, o_dgamma, o_dbeta");
        p.o_mean = d->o_mean; p.o_invstd = d->o_invstd; p.o_gamma = d->o_gamma; p.o_dgamma = d->o_dgamma; p.o_dbeta = d->o_dbeta;
    }
    if (d->epi != 0) {
        KPGNN_REQUIRE(d->out_slot != nullptr, "linear_bn: epi %d needs out_slot", d->epi);
        p.out_slot = d->out_slot;
    }
    if (d->epi == 2) {
        KPGNN_REQUIRE(d->e_x && d->e_mean && d->e_invstd && d->e_gamma && d->e_beta, "linear_bn: epi 2 needs e_x, e_mean, e_invstd, e_gamma, e_beta");
        p.e_x = d->e_x; p.e_mean = d->e_mean; p.e_invstd = d->e_invstd; p.e_gamma = d->e_gamma; p.e_beta = d->e_beta;
    }
    hipStream_t s = (hipStream_t)stream;
    // the bf16-split kernels (linear_bf3_fused.hip) where they apply: a batch that fills the chip, a row that fits their LDS
    // plan, the caller's scratch for the split copy of W
    if (d->math != KPGNN_MATH_F32 && d->workspace && (((uintptr_t)d->workspace) & 15) == 0 && d->N >= 4096 && d->I <= 104 &&
        d->workspace_bytes >= linear3_workspace_bytes(d->O, d->I, 1) && linear3_workspace_bytes(d->O, d->I, 1) > 0 &&
        (d->pro < 2 || d->xt != nullptr) && (d->w_transposed || (((uintptr_t)d->w) & 15) == 0)) {
        if (!d->w_split_ready) {
            const int rc = d->w_transposed ? linear3_split_w(d->w, 1, d->O, d->O, d->I, d->workspace, s)       // w [I, O]
                                           : linear3_split_w(d->w, d->I, 1, d->O, d->I, d->workspace, s);      // w [O, I]
            if (rc != KPGNN_OK) return rc;
        }
        return linear3_fused(p, d->pro, d->epi, (const uint4*)d->workspace, s);
    }
    switch (d->pro * 10 + d->epi) {
        case 0: return lin_launch_plain(p, s);
        case 1: return lin_launch_stats(p, s);
        case 10: return lin_launch_bn(p, s);
        case 11: return lin_launch_bn_stats(p, s);
        case 20: return lin_launch_bwd(p, s);
        case 22: return lin_launch_bwd_reduce(p, s);
        case 32: return lin_launch_bwd2_reduce(p, s);
        default: return fail(KPGNN_ELIMIT, "linear_bn: combination pro=%d epi=%d is not instantiated", d->pro, d->epi);
    }
}

extern "C" size_t kpgnn_stat_slot_bytes(int32_t C) {
    return C < 1 ? 0 : sizeof(double) * KPGNN_STAT_REPLICAS * 2 * (size_t)C;
}

extern "C" int kpgnn_stream_capture_id(kpgnn_stream_t stream, uint64_t* id) {
    KPGNN_REQUIRE(id != nullptr, "stream_capture_id: NULL id");
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long cid = 0;
    KPGNN_HIP_TRY(hipStreamGetCaptureInfo((hipStream_t)stream, &st, &cid));
    *id = st == hipStreamCaptureStatusActive ? (uint64_t)(cid ? cid : 1) : 0;
    return KPGNN_OK;
}
