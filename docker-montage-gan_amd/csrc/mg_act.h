// mg_act.h -- pointwise non-linearities of the bias_act family, shared by bias_act.hip and fused.hip.
//
// Layout of this file: one `Act<ID>` per activation id of the plugin API (1 = linear ... 9 = swish) exposing
//     fwd(t, a)          the function itself on the biased input t
//     slope(u, t, a)     f'   written in terms of the function VALUE u = f(t) where the API saves the output
//                        (`ref='y'` activations) or of the saved input t (`ref='x'`: swish)
//     curve(u, t, a)     f''  in the same variables
// and one driver, ba_eval(), that applies bias, gain, the incoming gradient and the clamp rule around them.
// What the API fixes (reference bias_act.cu:23-147 for the CUDA statement of the same contract):
//     order 0:  y = sat(f(x + b) * gain, +-clamp)
//     order 1:  dx = dy * f'(.) * gain                       where |y_saved| < clamp, else 0
//     order 2:  d2 = ddx * dy * f''(.) * gain                where |y_saved| < clamp, else 0
// with one-sided conventions at the kinks taken from the SAVED OUTPUT (u > 0 for relu/lrelu, u >= 0 for elu/selu).
#pragma once
#include "mg_common.h"

template <class acc_t> struct BaConst {
    acc_t alpha, gain, clamp;
};

template <int ID> struct Act;

template <> struct Act<1> {     // linear
    template <class R> static __device__ __forceinline__ R fwd(R t, R) { return t; }
    template <class R> static __device__ __forceinline__ R slope(R, R, R) { return R(1); }
    template <class R> static __device__ __forceinline__ R curve(R, R, R) { return R(0); }
};
template <> struct Act<2> {     // relu
    template <class R> static __device__ __forceinline__ R fwd(R t, R) { return t > R(0) ? t : R(0); }
    template <class R> static __device__ __forceinline__ R slope(R u, R, R) { return u > R(0) ? R(1) : R(0); }
    template <class R> static __device__ __forceinline__ R curve(R, R, R) { return R(0); }
};
template <> struct Act<3> {     // leaky relu, negative slope a
    template <class R> static __device__ __forceinline__ R fwd(R t, R a) { return t > R(0) ? t : t * a; }
    template <class R> static __device__ __forceinline__ R slope(R u, R, R a) { return u > R(0) ? R(1) : a; }
    template <class R> static __device__ __forceinline__ R curve(R, R, R) { return R(0); }
};
template <> struct Act<4> {     // tanh: f' = 1 - u^2, f'' = -2u f'
    template <class R> static __device__ __forceinline__ R fwd(R t, R) { return tanh(t); }
    template <class R> static __device__ __forceinline__ R slope(R u, R, R) { return R(1) - u * u; }
    template <class R> static __device__ __forceinline__ R curve(R u, R, R) { return (u * u - R(1)) * (u + u); }
};
template <> struct Act<5> {     // logistic: f' = u(1-u), f'' = f'(1-2u)
    template <class R> static __device__ __forceinline__ R fwd(R t, R) { return R(1) / (R(1) + exp(-t)); }       // exp overflow -> 1/inf = 0
    template <class R> static __device__ __forceinline__ R slope(R u, R, R) { return u - u * u; }
    template <class R> static __device__ __forceinline__ R curve(R u, R, R) { return (u - u * u) * (R(1) - (u + u)); }
};
template <> struct Act<6> {     // elu: on the negative branch u = e^t - 1, so f' = f'' = u + 1
    template <class R> static __device__ __forceinline__ R fwd(R t, R) { return t >= R(0) ? t : expm1(t); }
    template <class R> static __device__ __forceinline__ R slope(R u, R, R) { return u >= R(0) ? R(1) : u + R(1); }
    template <class R> static __device__ __forceinline__ R curve(R u, R, R) { return u >= R(0) ? R(0) : u + R(1); }
};
template <> struct Act<7> {     // selu = lam * elu_alpha; negative branch u = lam*alp*(e^t - 1), so f' = f'' = u + lam*alp
    static constexpr double LAM = 1.05070098735548049342, ALP = 1.67326324235437728482;
    template <class R> static __device__ __forceinline__ R fwd(R t, R) { return t >= R(0) ? R(LAM) * t : R(LAM * ALP) * expm1(t); }
    template <class R> static __device__ __forceinline__ R slope(R u, R, R) { return u >= R(0) ? R(LAM) : u + R(LAM * ALP); }
    template <class R> static __device__ __forceinline__ R curve(R u, R, R) { return u >= R(0) ? R(0) : u + R(LAM * ALP); }
};
template <> struct Act<8> {     // softplus u = log(1 + e^t): e^-u = 1 - sigmoid(t), hence f' = 1 - e^-u, f'' = e^-u (1 - e^-u)
    template <class R> static __device__ __forceinline__ R fwd(R t, R) { return (t > R(0) ? t : R(0)) + log1p(exp(-fabs(t))); }
    template <class R> static __device__ __forceinline__ R slope(R u, R, R) { return -expm1(-u); }
    template <class R> static __device__ __forceinline__ R curve(R u, R, R) { R q = exp(-u); return q - q * q; }
};
template <> struct Act<9> {     // swish t * s(t), derivatives from the saved INPUT t: f' = s (1 + t (1 - s)), f'' = s (1 - s) (2 + t (1 - 2s))
    template <class R> static __device__ __forceinline__ R sig(R t) { return R(1) / (R(1) + exp(-t)); }
    template <class R> static __device__ __forceinline__ R fwd(R t, R) { return t * sig(t); }
    template <class R> static __device__ __forceinline__ R slope(R, R t, R) { R s = sig(t); return s + t * (s - s * s); }
    template <class R> static __device__ __forceinline__ R curve(R, R t, R) { R s = sig(t); return (s - s * s) * (R(2) + t * (R(1) - (s + s))); }
};

// One element.  A = activation id, G = gradient order (0: x is the input; 1: x is the incoming gradient; 2: x is the
// gradient of the first-order result and dy the original incoming gradient).  xref / yref are the tensors the API saved
// (0 where it saved none); b is the bias that goes with x (order 0) or with xref (orders 1, 2).
template <class acc_t, int A>
__device__ __forceinline__ acc_t ba_eval(int G, acc_t x, acc_t b, acc_t xref, acc_t yref, acc_t dy,
                                         const BaConst<acc_t>& k)
{
    typedef Act<A> F;
    const acc_t lim = k.clamp;
    if (G == 0) {
        acc_t y = F::fwd(x + b, k.alpha) * (k.gain * dy);
        if (lim >= 0 && !(y > -lim && y < lim)) y = y >= 0 ? lim : -lim;       // saturate (a NaN lands on -clamp, as the API's CUDA statement does)
        return y;
    }
    const acc_t t = xref + b;
    acc_t u;                        // the function value before the gain
    if (A == 9) {                   // saved its input: rebuild the output for the clamp test
        yref = F::fwd(t, k.alpha) * k.gain;
        u = 0;
    } else if (A <= 3) {            // only the SIGN of u matters: gain's sign answers it without a division per element
        u = k.gain > 0 ? yref : (k.gain < 0 ? -yref : acc_t(0));
    } else {
        u = k.gain != 0 ? yref / k.gain : acc_t(0);
    }
    acc_t d = (G == 1) ? F::slope(u, t, k.alpha) : F::curve(u, t, k.alpha);
    acc_t y = x * d * (k.gain * dy);
    if (lim >= 0 && !(yref > -lim && yref < lim)) y = 0;
    return y;
}
