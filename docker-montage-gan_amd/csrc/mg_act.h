// mg_act.h -- the activation table shared by bias_act.hip and fused.hip.
#pragma once
#include "mg_common.h"

template <class acc_t> struct BaConst {
    acc_t alpha, gain, clamp;
};

// One element.  A = activation id 1..9, G = gradient order.
// Semantics per reference bias_act.cu:51-142.
template <class acc_t, int A>
__device__ __forceinline__ acc_t ba_eval(int G, acc_t x, acc_t b, acc_t xref, acc_t yref, acc_t dy,
                                         const BaConst<acc_t>& k)
{
    const acc_t one = 1, two = 2, expRange = 80, halfExpRange = 40;
    const acc_t seluScale = (acc_t)1.0507009873554804934193349852946;
    const acc_t seluAlpha = (acc_t)1.6732632423543772848170429916717;
    // yy = yref / gain (bias_act.cu:48).  relu/lrelu only ask whether yy > 0, which the sign of gain
    // answers without an IEEE division per element; linear never reads it.
    acc_t yy;
    if (A <= 3) yy = (k.gain > 0) ? yref : (k.gain < 0) ? -yref : 0;
    else        yy = (k.gain != 0) ? yref / k.gain : 0;
    acc_t y = 0;
    if (G == 0) x += b; else xref += b;

    if (A == 1) { if (G <= 1) y = x; }
    if (A == 2) { if (G == 0) y = (x > 0) ? x : 0; if (G == 1) y = (yy > 0) ? x : 0; }
    if (A == 3) { if (G == 0) y = (x > 0) ? x : x * k.alpha; if (G == 1) y = (yy > 0) ? x : x * k.alpha; }
    if (A == 4) {
        if (G == 0) { acc_t c = exp(x), d = one / c; y = (x < -expRange) ? -one : (x > expRange) ? one : (c - d) / (c + d); }
        if (G == 1) y = x * (one - yy * yy);
        if (G == 2) y = x * (one - yy * yy) * (-two * yy);
    }
    if (A == 5) {
        if (G == 0) y = (x < -expRange) ? 0 : one / (exp(-x) + one);
        if (G == 1) y = x * yy * (one - yy);
        if (G == 2) y = x * yy * (one - yy) * (one - two * yy);
    }
    if (A == 6) {
        if (G == 0) y = (x >= 0) ? x : exp(x) - one;
        if (G == 1) y = (yy >= 0) ? x : x * (yy + one);
        if (G == 2) y = (yy >= 0) ? 0 : x * (yy + one);
    }
    if (A == 7) {
        if (G == 0) y = (x >= 0) ? seluScale * x : (seluScale * seluAlpha) * (exp(x) - one);
        if (G == 1) y = (yy >= 0) ? x * seluScale : x * (yy + seluScale * seluAlpha);
        if (G == 2) y = (yy >= 0) ? 0 : x * (yy + seluScale * seluAlpha);
    }
    if (A == 8) {
        if (G == 0) y = (x > expRange) ? x : log(exp(x) + one);
        if (G == 1) y = x * (one - exp(-yy));
        if (G == 2) { acc_t c = exp(-yy); y = x * c * (one - c); }
    }
    if (A == 9) {
        if (G == 0) {
            y = (x < -expRange) ? 0 : x / (exp(-x) + one);
        } else {
            acc_t c = exp(xref), d = c + one;
            if (G == 1) y = (xref > halfExpRange) ? x : x * c * (xref + d) / (d * d);
            else        y = (xref > halfExpRange) ? 0 : x * c * (xref * (two - d) + two * d) / (d * d * d);
            yref = (xref < -expRange) ? 0 : xref / (exp(-xref) + one) * k.gain;
        }
    }
    y *= k.gain * dy;
    if (k.clamp >= 0) {
        if (G == 0) y = (y > -k.clamp & y < k.clamp) ? y : (y >= 0) ? k.clamp : -k.clamp;
        else        y = (yref > -k.clamp & yref < k.clamp) ? y : 0;
    }
    return y;
}

