/* det6d_rng.h — counter-based randomness of the device-side input producer (shared, like
 * det6d_math.h, by the HIP kernels and the CPU oracle so both draw the same numbers).
 *
 * The reference samples with numpy's global Mersenne Twister (np.random.choice / np.random.shuffle,
 * core/pcdet/datasets/processor/data_processor.py:145-178): inherently sequential and tied to the
 * DataLoader worker's state.  The build keeps the reference's SELECTION RULE and replaces the
 * generator by keyed bijections on [0, n): a random k-subset of n items is {r : perm_n(r) < k}, a
 * shuffle of n slots is slot -> perm_n(slot).  perm_n is a 4-round Feistel network over the smallest
 * even-bit domain >= n, cycle-walked back into [0, n).  Fully parallel, reproducible from
 * (seed, scene, purpose), and independent of thread / block geometry.
 */
#ifndef DET6D_RNG_H_
#define DET6D_RNG_H_

#include <stdint.h>

#ifndef D6_HD
#if defined(__HIPCC__)
#define D6_HD __host__ __device__ __forceinline__
#else
#define D6_HD static inline
#endif
#endif

/* 32-bit finaliser ("lowbias32") */
D6_HD uint32_t d6_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU;
  x ^= x >> 15; x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

/* key of one (seed, scene, purpose) stream */
D6_HD uint32_t d6_stream_key(uint64_t seed, uint32_t scene, uint32_t purpose) {
  uint32_t k = d6_mix32((uint32_t)seed ^ 0x9e3779b9U);
  k = d6_mix32(k ^ (uint32_t)(seed >> 32));
  k = d6_mix32(k ^ (scene * 0x85ebca6bU + purpose));
  return k;
}

/* keyed bijection of [0, n), n >= 1 */
D6_HD uint32_t d6_perm(uint32_t x, uint32_t n, uint32_t key) {
  if (n <= 1) return 0;
  uint32_t bits = 0;
  while (bits < 32 && ((uint64_t)1 << bits) < n) ++bits;
  const uint32_t h = (bits + 1) >> 1;          /* half width */
  const uint32_t mask = ((uint32_t)1 << h) - 1;
  do {
    uint32_t l = x >> h, r = x & mask;
    for (uint32_t round = 0; round < 4; ++round) {
      const uint32_t f = d6_mix32(r ^ (key + round * 0x9e3779b9U)) & mask;
      const uint32_t t = l ^ f;
      l = r;
      r = t;
    }
    x = (l << h) | r;
  } while (x >= n);
  return x;
}

/* uniform integer in [0, n) for the with-replacement draws */
D6_HD uint32_t d6_randint(uint32_t counter, uint32_t n, uint32_t key) {
  const uint32_t u = d6_mix32(d6_mix32(counter ^ key) + 0x68e31da4U);
  return (uint32_t)(((uint64_t)u * n) >> 32);
}

#endif /* DET6D_RNG_H_ */
