/*
 * montage_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C restatement of the reference's algorithms for the hot path
 * (SURVEY.md section 8a rows a1, a3, a9, a10).  Only tests/,
 * __graft_entry__.smoke() and the cpu_baseline leg of bench.py may load the
 * library built from this file; the product package never does, and its GPU
 * path raises when libmontage_hip.so is missing instead of coming here.
 *
 * Parity status: PINNED.  The reference ships no golden vectors (SURVEY.md
 * section 4), so tests/golden/make_golden.py imports the reference's own CPU
 * `ref` implementations in the authoring container and commits their
 * inputs/outputs under tests/golden/; tests/test_oracle_golden.py checks this
 * oracle against every one of them.
 *
 * Build: `make -C oracle`  ->  oracle/libmontage_oracle.so
 */
#include <math.h>
#include <float.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

/* floor(a / b) for b > 0 and any a (torch_utils/ops/upfirdn2d.cu:20-24). */
static int orc_floor_div(int a, int b)
{
    int q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) q--;
    return q;
}

static int orc_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Integer receptive field of output coordinate `o` along one axis
 * (torch_utils/ops/upfirdn2d.cu:43-48 for Y, 60-65 for X):
 *   mid    = o*down + up - 1 - pad0
 *   in0    = clamp(floor(mid / up), 0, in_size)
 *   count  = clamp(floor((mid + fsize) / up), 0, in_size) - in0
 *   f0     = mid + fsize - (in0 + 1)*up,  mirrored to fsize-1-f0 when flip
 * the filter index then advances by -up (flip: +up) per input sample (74-75). */
static void orc_axis(int o, int up, int down, int pad0, int fsize, int in_size, int flip,
                     int* in0, int* count, int* f0)
{
    int mid = o * down + up - 1 - pad0;
    int i0 = orc_clampi(orc_floor_div(mid, up), 0, in_size);
    int i1 = orc_clampi(orc_floor_div(mid + fsize, up), 0, in_size);
    int ff = mid + fsize - (i0 + 1) * up;
    if (flip) ff = fsize - 1 - ff;
    *in0 = i0; *count = i1 - i0; *f0 = ff;
}

/* table[o*3 + {0,1,2}] = {in0, count, f0}; the bit-exact part of upfirdn2d. */
void orc_upfirdn2d_index_table(int32_t* table, int out_size, int in_size, int up, int down,
                               int pad0, int fsize, int flip)
{
    for (int o = 0; o < out_size; o++) {
        int a, b, c;
        orc_axis(o, up, down, pad0, fsize, in_size, flip, &a, &b, &c);
        table[3 * o] = a; table[3 * o + 1] = b; table[3 * o + 2] = c;
    }
}

/* out = (in*up + pad0 + pad1 - fsize + down) / down  (upfirdn2d.cpp:32-33, C
 * integer division, truncating). */
int orc_upfirdn2d_out_size(int in_size, int up, int down, int pad0, int pad1, int fsize)
{
    return (in_size * up + pad0 + pad1 - fsize + down) / down;
}

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define REAL float
#define FN(name) CAT(name, _f32)
#define REXP expf
#define RLOG logf
#define RMAX FLT_MAX
#include "montage_oracle_body.inc"
#undef REAL
#undef FN
#undef REXP
#undef RLOG
#undef RMAX

#define REAL double
#define FN(name) CAT(name, _f64)
#define REXP exp
#define RLOG log
#define RMAX DBL_MAX
#include "montage_oracle_body.inc"
#undef REAL
#undef FN
#undef REXP
#undef RLOG
#undef RMAX
