// launchrate.hip -- how fast does the chip start one-wavefront workgroups?  (tools, not product)
//   hipcc --offload-arch=gfx950 -O3 tools/launchrate.hip -o /tmp/launchrate && /tmp/launchrate
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int LDS_WORDS, int SPIN>
__global__ void __launch_bounds__(64) k_empty(float *out, const float *in)
{
    __shared__ float s[LDS_WORDS > 0 ? LDS_WORDS : 1];
    float v = 0.0f;
    if (SPIN > 0) {
        v = in[blockIdx.x & 1023];
#pragma unroll 1
        for (int i = 0; i < SPIN; i++) v = v * 1.0001f + 0.5f;
    }
    if (LDS_WORDS > 0) { s[threadIdx.x] = v; __syncthreads(); v = s[63 - threadIdx.x]; }
    if (v == 12345.678f) out[blockIdx.x] = v;
}
template <int L, int S>
void run(const char *name, int blocks, int threads, float *out, float *in)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) k_empty<L, S><<<blocks, threads>>>(out, in);
    hipEventRecord(a);
    for (int i = 0; i < 10; i++) k_empty<L, S><<<blocks, threads>>>(out, in);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("%-34s blocks %7d x %3d thr: %8.1f us  = %6.1f workgroups/us\n", name, blocks, threads, ms * 1e3, blocks / (ms * 1e3));
}
int main()
{
    float *out, *in; hipMalloc(&out, 1 << 22); hipMalloc(&in, 4096); hipMemset(in, 0, 4096);
    run<0, 0>("empty", 262144, 64, out, in);
    run<800, 0>("3.2 KB LDS + barrier", 262144, 64, out, in);
    run<800, 1000>("3.2 KB LDS, 1000 dependent FMAs", 262144, 64, out, in);
    run<800, 4000>("3.2 KB LDS, 4000 dependent FMAs", 262144, 64, out, in);
    run<0, 0>("empty, 256-thread groups", 65536, 256, out, in);
    run<800, 4000>("256-thread, 4000 dependent FMAs", 65536, 256, out, in);
    return 0;
}
