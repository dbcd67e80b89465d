// bias_act.hip -- fused bias + activation + gain + clamp and its 1st/2nd order
// gradient kernels for gfx950.  Replaces bias_act_plugin.bias_act
// (reference torch_utils/ops/bias_act.cu:23-147, bias_act.cpp:32-90).
//
// HBM-bound streaming op: 16-byte vector loads/stores per lane, several vectors
// in flight per lane, grid-stride over chunks; the bias index (i / step_b) %
// size_b is one multiply-shift per 16-byte vector when step_b is a multiple of
// the vector width (NCHW planes), one 16-byte bias vector per lane when the bias
// dimension is the contiguous one and a multiple of the vector width (NHWC), per
// element otherwise (FC layers with odd widths).
#include "mg_common.h"
#include "mg_act.h"

namespace {

template <class T> struct BaArgs {
    const T* x; const T* b; const T* xref; const T* yref; const T* dy; T* y;
    int grad; float alpha, gain, clamp;
    uint32_t size_x, size_b;
    FastDiv step_div, sizeb_div;
};

template <class T> __device__ __forceinline__ uint32_t ba_bias_index(uint32_t i, const BaArgs<T>& p)
{
    uint32_t q = fd_div(i, p.step_div);
    return q - fd_div(q, p.sizeb_div) * p.size_b;
}

// VEC elements per lane per access (16 bytes, or 1 when pointers are unaligned);
// U accesses in flight per lane.  BVEC: 1 = bias index is uniform inside one vector, 2 = the vector's
// elements take consecutive biases (step_b == 1, size_b % VEC == 0: channels-last), 0 = per element.
// GT: gradient order known at compile time (0 or 1; -1 = read p.grad).  With GT fixed the loads that
// order never needs (xref/yref/dy at order 0; b/xref/dy for the 'y'-referenced activations at order 1)
// and their null checks disappear from the per-element code.
// One chunk of 256 * U vectors.  FULL: every lane of every access is inside the tensor (no bounds test); HXR / HY / HDY: which
// of the saved tensors this launch reads -- compile-time, so the chunk's loads issue back to back (tested at run time, each
// load sat in its own basic block behind `s_waitcnt vmcnt(0)`).
template <class T, int A, int VEC, int U, int BVEC, bool FULL, int RXR, int RY, int RDY, class P, class K>
__device__ __forceinline__ void ba_chunk(const P& p, const K& k, int G, uint32_t base, uint32_t nvec)
{
    typedef typename Acc<T>::type acc_t;
    struct alignas(sizeof(T) * VEC) V { T v[VEC]; };
    // 0 / 1: known at compile time; 2: whatever the pointer says (the generic instantiations)
    const bool HXR = RXR == 2 ? p.xref != nullptr : RXR != 0, HY = RY == 2 ? p.yref != nullptr : RY != 0, HDY = RDY == 2 ? p.dy != nullptr : RDY != 0;
    V vx[U], vxr[U], vyr[U], vdy[U];
    #pragma unroll
    for (int u = 0; u < U; u++) {
        uint32_t vi = base + u * 256u + threadIdx.x;
        if (FULL || vi < nvec) {
            vx[u] = ((const V*)p.x)[vi];
            if (HXR) vxr[u] = ((const V*)p.xref)[vi];
            if (HY) vyr[u] = ((const V*)p.yref)[vi];
            if (HDY) vdy[u] = ((const V*)p.dy)[vi];
        }
    }
    #pragma unroll
    for (int u = 0; u < U; u++) {
        uint32_t vi = base + u * 256u + threadIdx.x;
        if (!FULL && vi >= nvec) continue;
        acc_t bv = 0;
        V bvec;
        if (BVEC == 1 && p.b) bv = to_acc<T>(p.b[ba_bias_index(vi * VEC, p)]);
        if (BVEC == 2 && p.b) bvec = *(const V*)(p.b + ba_bias_index(vi * VEC, p));
        V out;
        #pragma unroll
        for (int e = 0; e < VEC; e++) {
            if (BVEC == 0 && p.b) bv = to_acc<T>(p.b[ba_bias_index(vi * VEC + e, p)]);
            if (BVEC == 2 && p.b) bv = to_acc<T>(bvec.v[e]);
            acc_t xr = HXR ? to_acc<T>(vxr[u].v[e]) : (acc_t)0;
            acc_t yr = HY ? to_acc<T>(vyr[u].v[e]) : (acc_t)0;
            acc_t dy = HDY ? to_acc<T>(vdy[u].v[e]) : (acc_t)1;
            out.v[e] = from_acc<T>(ba_eval<acc_t, A>(G, to_acc<T>(vx[u].v[e]), bv, xr, yr, dy, k));
        }
        ((V*)p.y)[vi] = out;
    }
}

template <class T, int A, int VEC, int U, int BVEC, int GT, bool FULL, class P, class K>
__device__ __forceinline__ void ba_chunk_by_reads(const P& p, const K& k, int G, uint32_t base, uint32_t nvec)
{
    if (GT == 0) ba_chunk<T, A, VEC, U, BVEC, FULL, 0, 0, 0>(p, k, G, base, nvec);
    else if (GT == 1 && A <= 8) {
        if (p.yref) ba_chunk<T, A, VEC, U, BVEC, FULL, 0, 1, 0>(p, k, G, base, nvec);       // (uniform)
        else ba_chunk<T, A, VEC, U, BVEC, FULL, 0, 0, 0>(p, k, G, base, nvec);
    } else ba_chunk<T, A, VEC, U, BVEC, FULL, 2, 2, 2>(p, k, G, base, nvec);
}

template <class T, int A, int VEC, int U, int BVEC, int GT>
__global__ void __launch_bounds__(256) bias_act_kernel(BaArgs<T> p0)
{
    typedef typename Acc<T>::type acc_t;
    BaArgs<T> p = p0;
    if (GT == 0) { p.xref = nullptr; p.yref = nullptr; p.dy = nullptr; }
    if (GT == 1) { p.dy = nullptr; if (A <= 8) { p.xref = nullptr; p.b = nullptr; } }
    const BaConst<acc_t> k = {(acc_t)p.alpha, (acc_t)p.gain, (acc_t)p.clamp};
    const int G = (GT >= 0) ? GT : p.grad;
    const uint32_t nvec = p.size_x / VEC;
    const uint32_t chunk = 256u * U;
    for (uint32_t base = blockIdx.x * chunk; base < nvec; base += gridDim.x * chunk) {
        if (GT < 0 || base + chunk > nvec) ba_chunk_by_reads<T, A, VEC, U, BVEC, GT, false>(p, k, G, base, nvec);
        else ba_chunk_by_reads<T, A, VEC, U, BVEC, GT, true>(p, k, G, base, nvec);
    }
    // tail (size_x not a multiple of VEC): last block, first lanes
    if (VEC > 1 && blockIdx.x == gridDim.x - 1) {
        uint32_t i = nvec * VEC + threadIdx.x;
        if (i < p.size_x) {
            acc_t bv = p.b ? to_acc<T>(p.b[ba_bias_index(i, p)]) : (acc_t)0;
            acc_t xr = p.xref ? to_acc<T>(p.xref[i]) : (acc_t)0;
            acc_t yr = p.yref ? to_acc<T>(p.yref[i]) : (acc_t)0;
            acc_t dy = p.dy ? to_acc<T>(p.dy[i]) : (acc_t)1;
            p.y[i] = from_acc<T>(ba_eval<acc_t, A>(G, to_acc<T>(p.x[i]), bv, xr, yr, dy, k));
        }
    }
}

template <class T, int A, int GT>
void launch_bias_act_g(const BaArgs<T>& p, bool aligned, int bvec, hipStream_t stream)
{
    constexpr int VEC = 16 / sizeof(T);
    constexpr int U = 4;
    const uint32_t max_blocks = 256u * 8u;
    if (aligned) {
        uint32_t nvec = p.size_x / VEC;
        uint32_t blocks = (nvec + 256u * U - 1) / (256u * U);
        blocks = blocks < 1 ? 1 : (blocks > max_blocks ? max_blocks : blocks);
        if (bvec == 1)      hipLaunchKernelGGL((bias_act_kernel<T, A, VEC, U, 1, GT>), dim3(blocks), dim3(256), 0, stream, p);
        else if (bvec == 2) hipLaunchKernelGGL((bias_act_kernel<T, A, VEC, U, 2, GT>), dim3(blocks), dim3(256), 0, stream, p);
        else                hipLaunchKernelGGL((bias_act_kernel<T, A, VEC, U, 0, GT>), dim3(blocks), dim3(256), 0, stream, p);
    } else {
        uint32_t blocks = (p.size_x + 256u * U - 1) / (256u * U);
        blocks = blocks < 1 ? 1 : (blocks > max_blocks ? max_blocks : blocks);
        hipLaunchKernelGGL((bias_act_kernel<T, A, 1, U, 1, GT>), dim3(blocks), dim3(256), 0, stream, p);
    }
}

template <class T, int A>
int launch_bias_act(const BaArgs<T>& p, bool aligned, int bvec, hipStream_t stream)
{
    // linear / relu / lrelu are what the networks run: orders 0 and 1 get their own code
    if (A <= 3 && p.grad == 0)      launch_bias_act_g<T, A, (A <= 3 ? 0 : -1)>(p, aligned, bvec, stream);
    else if (A <= 3 && p.grad == 1) launch_bias_act_g<T, A, (A <= 3 ? 1 : -1)>(p, aligned, bvec, stream);
    else                            launch_bias_act_g<T, A, -1>(p, aligned, bvec, stream);
    return mg_check_launch(MG_K_BIAS_ACT, "bias_act");
}

template <class T>
int dispatch_act(const BaArgs<T>& p, int act, bool aligned, int bvec, hipStream_t s)
{
    switch (act) {
    case 1: return launch_bias_act<T, 1>(p, aligned, bvec, s);
    case 2: return launch_bias_act<T, 2>(p, aligned, bvec, s);
    case 3: return launch_bias_act<T, 3>(p, aligned, bvec, s);
    case 4: return launch_bias_act<T, 4>(p, aligned, bvec, s);
    case 5: return launch_bias_act<T, 5>(p, aligned, bvec, s);
    case 6: return launch_bias_act<T, 6>(p, aligned, bvec, s);
    case 7: return launch_bias_act<T, 7>(p, aligned, bvec, s);
    case 8: return launch_bias_act<T, 8>(p, aligned, bvec, s);
    case 9: return launch_bias_act<T, 9>(p, aligned, bvec, s);
    }
    mg_set_error("no kernel found for the specified activation func");   // bias_act.cpp:81
    return MG_E_ARG;
}

} // namespace

extern "C" int mg_bias_act(const void* x, const void* b, const void* xref, const void* yref, const void* dy,
                           void* y, int dtype, int grad, int act, float alpha, float gain, float clamp,
                           int64_t size_x, int32_t size_b, int64_t step_b, void* stream)
{
    MG_REQUIRE(x && y, "x must reside on the device");
    MG_REQUIRE(size_x >= 0 && size_x <= INT32_MAX, "x is too large");                          // bias_act.cpp:40
    MG_REQUIRE(grad >= 0 && grad <= 2, "grad must be 0, 1 or 2");                               // bias_act.cpp:44
    MG_REQUIRE(b == nullptr || (size_b >= 1 && step_b >= 1), "b has wrong number of elements"); // bias_act.cpp:43
    if (size_x == 0) return MG_OK;
    int es = mg_dtype_size(dtype);
    MG_REQUIRE(es > 0, "unsupported dtype %d", dtype);
    uintptr_t ptrs = (uintptr_t)x | (uintptr_t)y | (uintptr_t)xref | (uintptr_t)yref | (uintptr_t)dy;
    bool aligned = (ptrs & 15) == 0;
    int vec = 16 / es;
    if (!b) { size_b = 1; step_b = 1; }
    // inside one aligned vector the bias index is constant iff step_b is a multiple of the vector width
    int bvec = ((b == nullptr) || (step_b % vec == 0)) ? 1 : 0;
    // channels-last: the bias dimension is the contiguous one -> one aligned bias vector per lane
    if (b && step_b == 1 && size_b % vec == 0 && (((uintptr_t)b) & 15) == 0) bvec = 2;
    if (!aligned) bvec = 1;   // VEC = 1
    int nin = 1 + (xref != nullptr) + (yref != nullptr) + (dy != nullptr);
    MgLaunchScope scope(MG_K_BIAS_ACT, (hipStream_t)stream, (double)size_x * es * (nin + 1));
    MG_DISPATCH_DTYPE(dtype, {
        BaArgs<T> p;
        p.x = (const T*)x; p.b = (const T*)b; p.xref = (const T*)xref; p.yref = (const T*)yref; p.dy = (const T*)dy;
        p.y = (T*)y; p.grad = grad; p.alpha = alpha; p.gain = gain; p.clamp = clamp;
        p.size_x = (uint32_t)size_x; p.size_b = (uint32_t)size_b;
        p.step_div = make_fastdiv((uint32_t)(step_b > INT32_MAX ? INT32_MAX : step_b));
        p.sizeb_div = make_fastdiv((uint32_t)size_b);
        return dispatch_act<T>(p, act, aligned, bvec, (hipStream_t)stream);
    });
    return MG_OK;
}
