// gg_hash.hip — 64-bit content hash of a host buffer, on all host threads (host code only; no kernel).
//
// The reference reads its (A, W) arrays on every call (networks.py:28,69); the engine keeps a device copy per pair, found
// by array identity AND a hash of the whole buffer, so that an in-place edit between two calls is new data (engine.py,
// DESIGN.md section 1b).  Hashing 134 MB of float64 W on one Python thread costs 17 ms per call (xxhash holds the GIL);
// here the buffer is cut into chunks hashed concurrently (the ctypes call has released the GIL) and the chunk digests are
// folded in chunk order.  The lane round is the XXH64 round (a bijection of the accumulator for a fixed lane and of the
// lane for a fixed accumulator), four accumulators per chunk, merged by bijective steps: a change confined to one 8-byte
// lane always changes the digest; anything else escapes with probability 2^-64.
#include <stdint.h>
#include <string.h>

#include <thread>
#include <vector>

#include "gg_common.h"

namespace {

constexpr uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL, P4 = 0x85EBCA77C2B2AE63ULL,
                   P5 = 0x27D4EB2F165667C5ULL;

inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t round1(uint64_t acc, uint64_t lane) { return rotl(acc + lane * P2, 31) * P1; }
inline uint64_t merge(uint64_t h, uint64_t acc) { return (h ^ round1(0, acc)) * P1 + P4; }

uint64_t hash_chunk(const uint8_t* p, size_t n, uint64_t seed) {
  uint64_t a0 = seed + P1 + P2, a1 = seed + P2, a2 = seed, a3 = seed - P1;
  size_t i = 0;
  for (; i + 32 <= n; i += 32) {
    uint64_t l[4];
    memcpy(l, p + i, 32);
    a0 = round1(a0, l[0]); a1 = round1(a1, l[1]); a2 = round1(a2, l[2]); a3 = round1(a3, l[3]);
  }
  uint64_t h = rotl(a0, 1) + rotl(a1, 7) + rotl(a2, 12) + rotl(a3, 18);
  h = merge(h, a0); h = merge(h, a1); h = merge(h, a2); h = merge(h, a3);
  h += (uint64_t)n;
  for (; i + 8 <= n; i += 8) {
    uint64_t l;
    memcpy(&l, p + i, 8);
    h = rotl(h ^ round1(0, l), 27) * P1 + P4;
  }
  for (; i < n; ++i) h = rotl(h ^ (p[i] * P5), 11) * P1;
  h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
  return h;
}

}  // namespace

int gg_hash64(const void* buf, size_t nbytes, int32_t threads, uint64_t* out) {
  if (!out || (!buf && nbytes)) return gg_fail(GG_ERR_INVALID, "gg_hash64: null argument");
  const uint8_t* p = static_cast<const uint8_t*>(buf);
  // fixed chunk size: the digest must not depend on the number of threads
  constexpr size_t CH = (size_t)4 << 20;
  const size_t nch = nbytes ? (nbytes + CH - 1) / CH : 1;
  std::vector<uint64_t> dig(nch);
  auto work = [&](size_t c0, size_t c1) {
    for (size_t c = c0; c < c1; ++c) {
      const size_t off = c * CH, len = off < nbytes ? (nbytes - off < CH ? nbytes - off : CH) : 0;
      dig[c] = hash_chunk(p + off, len, (uint64_t)c);
    }
  };
  int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
  if (nt < 1) nt = 1;
  if (nt > 32) nt = 32;
  if ((size_t)nt > nch) nt = (int)nch;
  if (nt <= 1) {
    work(0, nch);
  } else {
    std::vector<std::thread> th;
    const size_t per = (nch + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
      const size_t c0 = (size_t)t * per, c1 = c0 + per < nch ? c0 + per : nch;
      if (c0 < c1) th.emplace_back(work, c0, c1);
    }
    for (auto& t : th) t.join();
  }
  *out = hash_chunk(reinterpret_cast<const uint8_t*>(dig.data()), nch * 8, (uint64_t)nbytes);
  return GG_OK;
}
