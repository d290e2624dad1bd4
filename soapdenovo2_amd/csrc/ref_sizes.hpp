// ref_sizes.hpp -- the reference's k-mer-set size schedule (newhash.c:142-185, 340-455), shared by the host replay and the
// device layout of growable sets.
#pragma once
#include <math.h>
#include <stdint.h>

#include <vector>

namespace pg {

// ---- the reference's size schedule (newhash.c:142-185) ----------------------------------------------
// is_prime_kh tests odd divisors 3 <= i < (u64)sqrt((float)n) -- strict '<' and a float sqrt, so squares of
// primes pass; table sizes must come from exactly this function.
inline bool ref_is_prime(uint64_t n) {
    if (n < 4) return true;
    if ((n & 1) == 0) return false;
    const uint64_t lim = (uint64_t)sqrt((float)n);
    for (uint64_t i = 3; i < lim; i += 2)
        if (n % i == 0) return false;
    return true;
}
inline uint64_t ref_next_prime(uint64_t n) {
    if ((n & 1) == 0) n++;
    while (!ref_is_prime(n)) n += 2;
    return n;
}

// One stretch of a growable set's life at one size: the set has `size` slots while its keys n_old + 1 .. n_end arrive (count
// after the put); n_old keys were there when it grew to this size (0 for the first).  put_kmerset tests the growth before it
// probes (newhash.c:477): the put that would make count exceed max = (u64)((float)size * 0.77f) grows the set first.
struct GrowEpoch { uint64_t size, n_old, n_end; };
// n distinct keys in all; trailing_put: a duplicate put arrived after the last new key (it still runs the growth test)
inline std::vector<GrowEpoch> grow_schedule(uint64_t n, bool trailing_put, uint64_t init_size) {
    std::vector<GrowEpoch> out;
    const float lf = 0.77f;
    uint64_t size = init_size, have = 0;
    for (;;) {
        const uint64_t max = (uint64_t)((float)size * lf);
        const uint64_t n_end = n < max ? n : max;
        out.push_back(GrowEpoch{size, have, n_end});
        have = n_end;
        const bool more = n > max || (n == max && trailing_put);     // the next put (a new key, or the trailing duplicate) finds count + 1 > max
        if (!more) break;
        uint64_t nn = size;
        do {
            nn = (nn < 0xFFFFFFFULL) ? (nn << 1) : (nn + 0xFFFFFFULL);
            nn = ref_next_prime(nn);
        } while ((float)nn * lf < (float)(have + 1));
        size = nn;
        if (n == max && have == n) { out.push_back(GrowEpoch{size, have, have}); break; }   // grown by the trailing duplicate: no key arrives at this size
    }
    return out;
}

}  // namespace pg
