// K-heavy 1x1 layers of the tower on the 256 x 256 8-phase contraction (gemm_8phase.h; round 6): first-block
// reductions of res4 / res5, [conv3 | downsample] of res4 / res5 (the block input sampled at stride 2 is the second pixel source), res5's
// conv3 + residual.  Tunable G8, default OFF: measured in the tower (profiles/r06_layers_g8.txt against r06_layers_g8off.txt, one box) the
// five layers take 101.2 / 110.6 / 69.2 / 87.6 / 50.5 us per 256 frames against 101.4 / 108.8 / 70.9 / 93.1 / 49.6 for conv1x1_wide_kernel
// (fragment-ordered weights streamed per wave) - within +- 6 %, + 0.5 ... 1.9 % on the bench line: the K loop is 25 - 40 % faster
// (1.24 - 1.47 PFLOP/s at these shapes, profiles/r06_gemm_8phase_v2.txt) but at K = 512 ... 1536 the layer time is set by the store pass
// nothing overlaps.  Kept as the measured reference of what a plain-HIP K loop reaches on this chip, and as a tested alternative.
// Reads the plain K-major weights [Cout][K] of the packed arena.  Summation order differs from conv1x1_wide_kernel (k-steps of 32 on 16x16 tiles against
// 16 on 32x32): the two agree at the bf16 tolerance, not bit for bit.
#include "conv.h"
#include "gemm_8phase.h"

namespace ivosw {

bool conv1x1_g8_ok(const ConvArgs& a) {
    if (!(a.KH == 1 && a.KW == 1 && a.pad == 0 && a.stride == 1 && a.H == a.Ho && a.W == a.Wo)) return false;
    const int K = a.Cin + (a.x2 ? a.Cin2 : 0);
    if (a.Cout % 256 || a.Cin % 64 || K % 128 || K < tune_get("WIDE1X1_K", 384)) return false;
    if (a.x2) {
        const int hw = a.Ho * a.Wo;
        // 64 tile rows further must be a whole number of raster rows of one frame, or of frames
        if (!((hw % 64 == 0 && 64 % a.Wo == 0) || 64 % hw == 0) || a.Cin2 % 64) return false;
        if ((size_t)a.B * a.H2 * a.W2 * a.Cin2 * 2 >= 0xfffffff0ull) return false;
    }
    // like conv1x1_wide_ok: the rule looks at the layer shape only, never at the batch (a frame's result must not depend on its chunk)
    return a.Ho * a.Wo * (a.Cout / 256) >= 256;
}

void launch_conv1x1_g8(const ConvArgs& a, hipStream_t st) {
    G8Args g{};
    g.A = static_cast<const bf16_t*>(a.x); g.B = static_cast<const bf16_t*>(a.w); g.bias = a.bias; g.C = static_cast<bf16_t*>(a.y);
    g.R = static_cast<const bf16_t*>(a.res);
    g.M = a.B * a.Ho * a.Wo; g.N = a.Cout; g.K1 = a.Cin; g.K2 = a.x2 ? a.Cin2 : 0; g.K = g.K1 + g.K2; g.ldc = a.Cout; g.relu = a.relu;
    g.A2 = static_cast<const bf16_t*>(a.x2); g.Ho = a.Ho; g.Wo = a.Wo; g.H2 = a.H2; g.W2 = a.W2; g.stride2 = a.stride2;
    g.rev = a.rev; g.ts = nullptr;
    void* tok = prof_begin(a, 2, st);
    const int grid = ((g.M + 255) / 256) * (g.N / 256);
    if (g.R) hipLaunchKernelGGL((gemm_8phase_kernel<0, true>), dim3(grid), dim3(512), 0, st, g);
    else hipLaunchKernelGGL((gemm_8phase_kernel<0, false>), dim3(grid), dim3(512), 0, st, g);
    prof_end(tok, st);
}

}  // namespace ivosw
