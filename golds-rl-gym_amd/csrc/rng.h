// Counter-based generator of the build (NOT the reference's MT19937): Philox-4x32-10,
// key = 64-bit seed, counter = (index, global env id, episode, stream).  CPU restatement used
// by the tests: oracle/oracle.py:rng_block / u01_pair / normal_pair.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace grl {

struct u32x4 { uint32_t v[4]; };

__device__ __forceinline__ u32x4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    u32x4 o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

__device__ __forceinline__ u32x4 rng_block(uint64_t seed, uint32_t env_id, uint32_t episode, uint32_t stream, uint32_t counter) {
    return philox4x32(counter, env_id, episode, stream, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// two float64 uniforms in [0,1): 53 high bits of each 64-bit half
__device__ __forceinline__ void u01_pair(const u32x4 &b, double &u0, double &u1) {
    uint64_t w0 = ((uint64_t)b.v[1] << 32) | b.v[0];
    uint64_t w1 = ((uint64_t)b.v[3] << 32) | b.v[2];
    u0 = (double)(w0 >> 11) * (1.0 / 9007199254740992.0);
    u1 = (double)(w1 >> 11) * (1.0 / 9007199254740992.0);
}

// Box-Muller on (1-u0, u1) in float64
__device__ __forceinline__ void normal_pair(const u32x4 &b, double &n0, double &n1) {
    double u0, u1;
    u01_pair(b, u0, u1);
    double r = sqrt(-2.0 * log(1.0 - u0));
    double th = 2.0 * 3.141592653589793 * u1;
    double s, c;
    sincos(th, &s, &c);
    n0 = r * c;
    n1 = r * s;
}

}  // namespace grl
