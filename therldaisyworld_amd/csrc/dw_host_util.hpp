// dw_host_util.hpp — host-side helpers of the C ABI that carry no HIP types, so that the CPU test suite can
// compile and exercise them with g++ (tests/test_abi_and_host.py).
#pragma once
#include <cstddef>

namespace dw {

// Two device buffers that are only ever used together (the two planes of an un-quantised state) are allocated
// ALL OR NOTHING: when the second allocation fails the first is given back and both pointers are null again, so
// that a later call sees "not allocated" and retries (or reports the failure again) instead of finding one
// plane and launching a kernel on a null second one.  `alloc(void**, size_t)` and `release(void*)` return 0 on
// success; the first failing code is returned.
template <class Alloc, class Release>
int alloc_pair_all_or_nothing(void** a, void** b, size_t bytes, Alloc alloc, Release release) {
    if (*a && *b) return 0;
    if (*a) { (void)release(*a); *a = nullptr; }                // left over from a partial failure of an older build
    if (*b) { (void)release(*b); *b = nullptr; }
    int rc = alloc(a, bytes);
    if (rc != 0) { *a = nullptr; return rc; }
    rc = alloc(b, bytes);
    if (rc != 0) {
        (void)release(*a);
        *a = nullptr;
        *b = nullptr;
        return rc;
    }
    return 0;
}

}  // namespace dw
