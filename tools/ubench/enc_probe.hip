// Phase timeline of enc_fused_kernel (s_memtime stamps per workgroup): build with
//   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 -DIVOSW_FUSED_PROBE -I../../ivos-w_amd/csrc -o enc_probe enc_probe.hip
#include "brain_fused.h"
#include <vector>
using namespace ivosw;
namespace ivosw { void set_error(const char*, ...) {} }
int main() {
    const int B = 128, T = 25, rows = B * T;
    float *prm, *x, *gx, *a1, *e;
    hipMalloc(&prm, 181000 * 4); hipMalloc(&x, 2 * rows * 2 * 4);
    hipMalloc(&gx, (size_t)3 * rows * 512 * 4); hipMalloc(&a1, (size_t)3 * rows * 128 * 4); hipMalloc(&e, (size_t)3 * rows * 128 * 4);
    std::vector<float> h(181000);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.01f * ((int)(i * 2654435761u % 201) - 100);
    hipMemcpy(prm, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(x, h.data(), 2 * rows * 2 * 4, hipMemcpyHostToDevice);
    EncGroup g{};
    g.j[0] = EncJob{prm, x, x + rows * 2, gx, a1, e, rows, 2 * rows, rows};
    g.j[1] = EncJob{prm, x, x, gx + (size_t)2 * rows * 512, a1 + (size_t)2 * rows * 128, e + (size_t)2 * rows * 128, rows, rows, rows};
    const int t0 = (2 * rows + FM - 1) / FM, t1 = (rows + FM - 1) / FM;
    g.first1 = t0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 3; ++pass) {
        hipEventRecord(e0);
        for (int r = 0; r < 100; ++r) hipLaunchKernelGGL(enc_fused_kernel, dim3(t0 + t1), dim3(256), 0, 0, g, 0, 256, 384, 16768, 16896);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        static unsigned long long hp[512][8];
        hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_fused_probe), sizeof(hp));
        double d[6] = {0}, lo = 1e30, hi = 0;
        for (int w = 0; w < t0 + t1; ++w) {
            for (int i = 1; i < 6; ++i) d[i] += (double)(hp[w][i] - hp[w][i - 1]);
            lo = std::min(lo, (double)hp[w][0]); hi = std::max(hi, (double)hp[w][5]);
        }
        printf("launch-to-launch %.2f us; ticks per phase (avg over %d workgroups): loads+sync %.0f  fc1 %.0f  fc2 %.0f  gates %.0f  stores %.0f; first start .. last end %.0f ticks\n",
               ms * 10, t0 + t1, d[1] / (t0 + t1), d[2] / (t0 + t1), d[3] / (t0 + t1), d[4] / (t0 + t1), d[5] / (t0 + t1), hi - lo);
    }
    return 0;
}
