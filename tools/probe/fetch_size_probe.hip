// Hardware probe (developer tool): what rocprofv3's FETCH_SIZE counter reports for a streaming read of a KNOWN number of bytes as a
// function of the per-lane load width (4 / 8 / 16 bytes: buffer_load_dword / dwordx2 / dwordx4, fully coalesced).  The guide
// (MI355X_MICROARCH.md, HBM) prescribes doubling FETCH_SIZE for wide coalesced reads on gfx950 because 128-byte requests are tallied
// as 64 bytes; that was calibrated on 16-byte loads.  bench.py's hbm_traffic() applies the factor measured here per kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/fetch_size_probe.hip -o tools/probe/fetch_size_probe
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- tools/probe/fetch_size_probe
// The buffer (1 GiB) is far larger than L2 + MALL, every kernel reads it exactly once: expected FETCH_SIZE = 1048576 KiB x factor^-1.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WIDTH>
__global__ void __launch_bounds__(256) k_read(const float* __restrict__ p, float* __restrict__ out, long long n_floats) {
    const long long stride = (long long)gridDim.x * 256 * WIDTH;
    float acc = 0.0f;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * WIDTH; i < n_floats; i += stride) {
        if (WIDTH == 1) acc += p[i];
        if (WIDTH == 2) { const f32x2 v = *reinterpret_cast<const f32x2*>(p + i); acc += v.x + v.y; }
        if (WIDTH == 4) { const f32x4 v = *reinterpret_cast<const f32x4*>(p + i); acc += (v.x + v.y) + (v.z + v.w); }
    }
    if (acc == 123.456f) out[0] = acc;
}
// the staging pattern of the split conv kernels: one dword per lane, consecutive lanes = consecutive voxels of an 18-wide haloed row
// (rows are 16 + 2 voxels: every third lane group starts a new, non-contiguous 72-byte run)
__global__ void __launch_bounds__(256) k_read_rows18(const float* __restrict__ p, float* __restrict__ out, long long n_floats, int W) {
    const long long nrow = n_floats / W;
    float acc = 0.0f;
    for (long long r = blockIdx.x; r < nrow; r += gridDim.x) {
        // 14 tiles of 16 columns per 224-wide row, each fetched with its one-voxel halo: 18 lanes per tile
        const int t = threadIdx.x / 18, l = threadIdx.x - 18 * t;
        if (t < W / 16) {
            const int w = t * 16 - 1 + l;
            if (w >= 0 && w < W) acc += p[r * W + w];
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    const long long n = 1ll << 28;                 // 1 GiB of floats
    float* p; float* out;
    (void)hipMalloc(&p, n * 4); (void)hipMalloc(&out, 4);
    (void)hipMemset(p, 0, n * 4);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_read<1>, dim3(8192), dim3(256), 0, 0, p, out, n);
        hipLaunchKernelGGL(k_read<2>, dim3(8192), dim3(256), 0, 0, p, out, n);
        hipLaunchKernelGGL(k_read<4>, dim3(8192), dim3(256), 0, 0, p, out, n);
        hipLaunchKernelGGL(k_read_rows18, dim3(8192), dim3(256), 0, 0, p, out, n, 224);
    }
    (void)hipDeviceSynchronize();
    printf("read %lld bytes per kernel launch (k_read_rows18: the same bytes + 2/16 halo re-reads served by the caches)\n", n * 4);
    return 0;
}
