// Microbenchmark: do partial writes of one cache line by workgroups on different XCDs merge?  (Each XCD has its own L2;
// a line written in part by two of them is dirty in both.)  Block b writes its share of every line of a buffer — halves
// (64 B each), then interleaved 16-byte, 8-byte and 4-byte pieces — and the host checks every byte after the kernel.
// Build: hipcc --offload-arch=gfx950 -O2 -o build/ubench_line_sharing tools/ubench_line_sharing.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void write_share(uint32_t* buf, int n_lines, int piece_words, int n_blocks, uint32_t* xcc) {
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        uint32_t id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[b] = id & 0xf;
    }
    const int words_per_line = 32;
    for (int rep = 0; rep < 4; ++rep)  // rewrite a few times: lines get evicted and refetched in between
        for (int i = threadIdx.x; i < n_lines * words_per_line; i += blockDim.x) {
            const int w = i % words_per_line;
            const int owner = (w / piece_words) % n_blocks;
            if (owner == b) buf[i] = 0x1000000u * (uint32_t)(b + 1) + (uint32_t)(i & 0xffffff);
        }
    __threadfence();
}

int main() {
    const int n_lines = 1 << 16;  // 8 MB: more than one L2
    uint32_t *d, *dx;
    hipMalloc(&d, (size_t)n_lines * 128);
    hipMalloc(&dx, 64 * 4);
    std::vector<uint32_t> h((size_t)n_lines * 32);
    for (int n_blocks : {2, 8})
        for (int piece : {16, 4, 2, 1}) {
            hipMemset(d, 0, (size_t)n_lines * 128);
            hipLaunchKernelGGL(write_share, dim3(n_blocks), dim3(1024), 0, 0, d, n_lines, piece, n_blocks, dx);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
            uint32_t x[8];
            hipMemcpy(x, dx, n_blocks * 4, hipMemcpyDeviceToHost);
            size_t bad = 0;
            for (size_t i = 0; i < h.size(); ++i) {
                const int w = (int)(i % 32);
                const int owner = (w / piece) % n_blocks;
                if (h[i] != 0x1000000u * (uint32_t)(owner + 1) + (uint32_t)(i & 0xffffff)) ++bad;
            }
            printf("blocks %d (XCCs", n_blocks);
            for (int q = 0; q < n_blocks; ++q) printf(" %u", x[q]);
            printf(") piece %3d B: %zu wrong words of %zu\n", piece * 4, bad, h.size());
        }
    return 0;
}
