// numpy-compatible integer/float draws for bit-exact sampled indices.
//
// The reference draws every replay index through numpy.random.default_rng
// (embodied/core/selectors.py:34,42,240,305; 212,216), i.e. SeedSequence ->
// PCG64 (XSL-RR 128/64) -> Generator.integers / random / choice.  numpy is a
// third-party dependency of the reference (requirements.txt pins numpy<2; 2.2
// in this image); this header restates its published algorithms so the host
// index core can reproduce the same streams without Python.  Checked against
// numpy itself in tests/test_host_rng.py.
#pragma once

#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace emb {

// SeedSequence (numpy/random/bit_generator.pyx): O'Neill's seed_seq_fe variant.
class SeedSequence {
 public:
  explicit SeedSequence(const std::vector<uint32_t>& entropy) {
    uint32_t hc = kInitA;
    auto hashmix = [&hc](uint32_t v) {
      v ^= hc;
      hc *= kMultA;
      v *= hc;
      v ^= v >> kShift;
      return v;
    };
    auto mix = [](uint32_t x, uint32_t y) {
      uint32_t r = kMixL * x - kMixR * y;
      r ^= r >> kShift;
      return r;
    };
    for (size_t i = 0; i < 4; ++i)
      pool_[i] = hashmix(i < entropy.size() ? entropy[i] : 0u);
    for (size_t src = 0; src < 4; ++src)
      for (size_t dst = 0; dst < 4; ++dst)
        if (src != dst) pool_[dst] = mix(pool_[dst], hashmix(pool_[src]));
    for (size_t src = 4; src < entropy.size(); ++src)
      for (size_t dst = 0; dst < 4; ++dst)
        pool_[dst] = mix(pool_[dst], hashmix(entropy[src]));
  }

  // Python int seed -> little-endian u32 words (0 -> one zero word).
  static std::vector<uint32_t> words_of(uint64_t seed) {
    std::vector<uint32_t> w;
    w.push_back(static_cast<uint32_t>(seed));
    if (seed >> 32) w.push_back(static_cast<uint32_t>(seed >> 32));
    return w;
  }

  void generate_u64(uint64_t* out, int n) const {
    uint32_t hc = kInitB;
    for (int i = 0; i < 2 * n; ++i) {
      uint32_t v = pool_[i % 4];
      v ^= hc;
      hc *= kMultB;
      v *= hc;
      v ^= v >> kShift;
      if (i % 2 == 0) out[i / 2] = v;
      else out[i / 2] |= static_cast<uint64_t>(v) << 32;
    }
  }

 private:
  static constexpr uint32_t kInitA = 0x43b0d7e5u, kMultA = 0x931e8875u;
  static constexpr uint32_t kInitB = 0x8b51f9ddu, kMultB = 0x58f38dedu;
  static constexpr uint32_t kMixL = 0xca01f9ddu, kMixR = 0x4973f715u;
  static constexpr int kShift = 16;
  uint32_t pool_[4];
};

// PCG64 bit generator + the Generator methods the reference uses.
class NpRandom {
 public:
  explicit NpRandom(uint64_t seed) : NpRandom(SeedSequence::words_of(seed)) {}

  explicit NpRandom(const std::vector<uint32_t>& entropy) {
    uint64_t s[4];
    SeedSequence(entropy).generate_u64(s, 4);
    const u128 initstate = (static_cast<u128>(s[0]) << 64) | s[1];
    const u128 initseq = (static_cast<u128>(s[2]) << 64) | s[3];
    state_ = 0;
    inc_ = (initseq << 1) | 1;
    step();
    state_ += initstate;
    step();
  }

  uint64_t next_u64() {
    step();
    const uint64_t hi = static_cast<uint64_t>(state_ >> 64);
    const uint64_t lo = static_cast<uint64_t>(state_);
    const uint64_t x = hi ^ lo;
    const unsigned rot = static_cast<unsigned>(state_ >> 122);
    return (x >> rot) | (x << ((64 - rot) & 63));
  }

  // Low half first, high half buffered for the next call.
  uint32_t next_u32() {
    if (has32_) {
      has32_ = false;
      return buf32_;
    }
    const uint64_t v = next_u64();
    has32_ = true;
    buf32_ = static_cast<uint32_t>(v >> 32);
    return static_cast<uint32_t>(v);
  }

  // Generator.random(): 53-bit mantissa; does not touch the u32 buffer.
  double random() { return (next_u64() >> 11) * (1.0 / 9007199254740992.0); }

  // Generator.integers(0, n) scalar, default int64, endpoint=False.
  int64_t integers(int64_t n) {
    const uint64_t rng = static_cast<uint64_t>(n) - 1;  // inclusive range
    if (rng == 0) return 0;                              // no draw
    if (rng <= 0xFFFFFFFFull) {
      if (rng == 0xFFFFFFFFull) return next_u32();
      const uint32_t excl = static_cast<uint32_t>(rng) + 1;
      uint64_t m = static_cast<uint64_t>(next_u32()) * excl;
      uint32_t left = static_cast<uint32_t>(m);
      if (left < excl) {
        const uint32_t thr = (0xFFFFFFFFu - static_cast<uint32_t>(rng)) % excl;
        while (left < thr) {
          m = static_cast<uint64_t>(next_u32()) * excl;
          left = static_cast<uint32_t>(m);
        }
      }
      return static_cast<int64_t>(m >> 32);
    }
    if (rng == ~0ull) return static_cast<int64_t>(next_u64());
    const uint64_t excl = rng + 1;
    u128 m = static_cast<u128>(next_u64()) * excl;
    uint64_t left = static_cast<uint64_t>(m);
    if (left < excl) {
      const uint64_t thr = (~0ull - rng) % excl;
      while (left < thr) {
        m = static_cast<u128>(next_u64()) * excl;
        left = static_cast<uint64_t>(m);
      }
    }
    return static_cast<int64_t>(m >> 64);
  }

  // Generator.choice(k, p=probs): cdf = cumsum(p); cdf /= cdf[-1];
  // searchsorted(cdf, random(), side='right').  `cdf` is scratch of size k.
  int choice(const double* probs, int k, double* cdf) {
    // The checks Generator.choice makes before it draws (same messages): a
    // sample tree whose masses went NaN ends here, as it does in the reference.
    double check = 0.0;
    bool negative = false;
    for (int i = 0; i < k; ++i) {
      check += probs[i];
      negative = negative || probs[i] < 0;
    }
    if (check != check) throw std::invalid_argument("probabilities contain NaN");
    if (negative) throw std::invalid_argument("probabilities are not non-negative");
    if (std::fabs(check - 1.0) > 1.4901161193847656e-08)
      throw std::invalid_argument("probabilities do not sum to 1");
    double acc = 0.0;
    for (int i = 0; i < k; ++i) {
      acc = (i == 0) ? probs[0] : acc + probs[i];
      cdf[i] = acc;
    }
    const double last = cdf[k - 1];
    for (int i = 0; i < k; ++i) cdf[i] /= last;
    const double u = random();
    int lo = 0, hi = k;
    while (lo < hi) {
      const int mid = lo + (hi - lo) / 2;
      if (u < cdf[mid]) hi = mid;
      else lo = mid + 1;
    }
    return lo;
  }

 private:
  using u128 = unsigned __int128;
  void step() {
    const u128 mult =
        (static_cast<u128>(0x2360ED051FC65DA4ull) << 64) | 0x4385DF649FCCF645ull;
    state_ = state_ * mult + inc_;
  }
  u128 state_, inc_;
  bool has32_ = false;
  uint32_t buf32_ = 0;
};

// numpy's pairwise float64 sum (numpy/_core/src/umath/loops_utils.h.src), as
// used by ndarray.sum() on a contiguous 1-D array.
inline double np_pairwise_sum(const double* a, int64_t n) {
  if (n < 8) {
    double res = 0.0;
    for (int64_t i = 0; i < n; ++i) res += a[i];
    return res;
  }
  if (n <= 128) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int64_t i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int64_t n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

}  // namespace emb
