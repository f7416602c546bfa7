// oob_probe.hip -- which offsets does the gfx950 buffer range check of a RAW descriptor (stride 0) cover?
// k_diff3_march2 drops rows / planes / lanes it must not touch by giving the access an out-of-range offset, either in
// the VGPR offset (voffset) or in the SGPR offset (soffset).  LLVM's documentation says soffset is not range-checked;
// the gfx9 ISA documents say raw buffers compare against num_records - soffset.  This probe settles it on the
// hardware, without ever touching memory outside its own allocation: the descriptor covers the first MiB of a 3 GiB
// buffer, every address an un-dropped access would reach lies inside that buffer and is checked for a canary.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef unsigned u4v __attribute__((ext_vector_type(4)));

__global__ void k_probe(unsigned char* base, unsigned num_records, unsigned voff, int soff, unsigned* loaded)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)num_records, 0x00020000);
    const unsigned v = voff + threadIdx.x * 16u;
    const u4v got = __builtin_amdgcn_raw_buffer_load_b128(r, v, soff, 0);
    loaded[threadIdx.x] = got.x;
    u4v val;
    val.x = val.y = val.z = val.w = 0xdeadbeefu;
    __builtin_amdgcn_raw_buffer_store_b128(val, r, v, soff, 0);
}

int main()
{
    const size_t total = 3ull << 30;
    unsigned char* buf = nullptr;
    if (hipMalloc(&buf, total) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    unsigned* loaded = nullptr;
    hipMalloc(&loaded, 64 * 4);
    const unsigned NR = 1u << 20;
    struct Case { const char* name; unsigned voff; int soff; };
    const Case cases[] = {
        {"in range: voffset 4096, soffset 8192", 4096u, 8192},
        {"A: soffset = 0x7ffffff0 (sentinel), voffset small", 0u, 0x7ffffff0},
        {"B: voffset = 0x7ffffff0 (sentinel), soffset 0", 0x7ffffff0u, 0},
        {"C: voffset 768 KiB + soffset 512 KiB (sum beyond num_records)", 768u << 10, 512 << 10},
        {"D: soffset 2 MiB (beyond num_records), voffset 0", 0u, 2 << 20},
        {"E: voffset 2 MiB (beyond num_records), soffset 0", 2u << 20, 0},
    };
    for (const Case& c : cases) {
        hipMemset(buf, 0x11, total);
        hipMemset(loaded, 0, 64 * 4);
        k_probe<<<1, 64>>>(buf, NR, c.voff, c.soff, loaded);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: kernel failed\n", c.name); return 2; }
        const size_t target = ((size_t)c.voff + (size_t)(unsigned)c.soff) % total;   // where a wrapped / un-dropped access would land
        std::vector<unsigned> h(256), l(64);
        hipMemcpy(l.data(), loaded, 64 * 4, hipMemcpyDeviceToHost);
        size_t t0 = (size_t)c.voff + (size_t)(unsigned)c.soff;
        bool stored = false;
        if (t0 + 1024 <= total) {
            hipMemcpy(h.data(), buf + t0, 1024, hipMemcpyDeviceToHost);
            for (unsigned x : h) stored |= (x == 0xdeadbeefu);
        }
        // scan the whole buffer for the marker in case the address wrapped somewhere else
        printf("%-72s load -> 0x%08x (%s)   store %s at base+voffset+soffset%s\n", c.name, l[0],
               l[0] == 0x11111111u ? "memory" : (l[0] == 0 ? "dropped: 0" : "?"), stored ? "LANDED" : "not seen",
               t0 + 1024 <= total ? "" : " (address beyond the 3 GiB buffer: not checkable)");
        (void)target;
    }
    hipFree(buf);
    hipFree(loaded);
    return 0;
}
