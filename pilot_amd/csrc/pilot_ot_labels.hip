// Host side of the pre-pass: label columns -> integer codes numbered in order of first appearance.
//
// The reference walks the annotation frame with pandas: annot.cell_type.unique() / annot.sampleID.unique() give the
// first-appearance order (pilotpy/tools/Trajectory.py:402, :412) and one boolean mask per label selects the rows
// (:405-425).  The device pre-pass needs the same partition as int32 codes.  pandas' own factorize is a single-threaded
// hash of 1.8 M values that holds the interpreter lock (4.6 + 5.6 ms for the two columns of BASELINE config 3 on the
// GPU box's host, a quarter of the whole tl.wasserstein_distance call); the labels arrive as fixed-width integers
// already -- the codes of a pandas Categorical, or the object pointers of an object column, of which a cohort holds a few
// hundred distinct values -- so the pass is: a last-value check, a small table, n_threads slices side by side, and the
// slices' first-appearance lists merged in slice order.  No device work, no HIP call.
#include <atomic>
#include <cstdint>
#include <cstring>
#include <system_error>
#include <thread>
#include <vector>

#include "abi_common.hpp"

namespace {

using pilot::abi_fail;

struct IdTable {                    // open addressing, 64-bit ids -> dense code; starts small, doubles at a quarter full
    std::vector<uint64_t> key;
    std::vector<int> val;
    uint64_t mask = 0;
    size_t used = 0;
    void init() {
        key.assign(1024, 0);
        val.assign(1024, -1);
        mask = 1023; used = 0;
    }
    static inline uint64_t mix(uint64_t x) {
        x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 29;
        return x;
    }
    // the code of `id`, or -1 with *slot = where it would go
    inline int find(uint64_t id, size_t *slot) const {
        size_t i = (size_t)(mix(id) & mask);
        while (val[i] >= 0) {
            if (key[i] == id) return val[i];
            i = (i + 1) & mask;
        }
        *slot = i;
        return -1;
    }
    // `slot` as find() left it (no insertion in between)
    inline void put(size_t slot, uint64_t id, int code) {
        key[slot] = id; val[slot] = code;
        if (++used * 4 > key.size()) grow();
    }
    void grow() {
        std::vector<uint64_t> k2(key.size() * 2, 0);
        std::vector<int> v2(key.size() * 2, -1);
        const uint64_t m2 = k2.size() - 1;
        for (size_t i = 0; i < key.size(); ++i) {
            if (val[i] < 0) continue;
            size_t j = (size_t)(mix(key[i]) & m2);
            while (v2[j] >= 0) j = (j + 1) & m2;
            k2[j] = key[i]; v2[j] = val[i];
        }
        key.swap(k2); val.swap(v2); mask = m2;
    }
};

struct Slice {
    long long begin = 0, end = 0;
    std::vector<uint64_t> ids;          // distinct ids in order of first appearance inside the slice
    std::vector<long long> first;       // row of that first appearance
    std::vector<int> remap;             // local code -> global code
    bool overflow = false;
};

template <typename T> inline bool id_missing(T v);
template <> inline bool id_missing<int8_t>(int8_t v) { return v < 0; }
template <> inline bool id_missing<int16_t>(int16_t v) { return v < 0; }
template <> inline bool id_missing<int32_t>(int32_t v) { return v < 0; }
template <> inline bool id_missing<uint64_t>(uint64_t v) { return v == 0; }

// one slice: local first-appearance codes into codes[begin, end)
template <typename T>
void encode_slice(const T *ids, Slice &s, int max_uniques, int *codes) {
    constexpr bool DIRECT = sizeof(T) <= 2;            // 1- and 2-byte ids index a table directly
    std::vector<int> direct;
    IdTable table;
    if (DIRECT) direct.assign((size_t)1 << (8 * sizeof(T)), -1);
    else table.init();
    T last = T(0);
    int last_code = -2;                                 // (-2: no last value yet)
    for (long long r = s.begin; r < s.end; ++r) {
        const T v = ids[r];
        if (last_code != -2 && v == last) { codes[r] = last_code; continue; }
        int code;
        if (id_missing<T>(v)) {
            code = -1;
        } else if (DIRECT) {
            int &slot = direct[(size_t)(typename std::make_unsigned<T>::type)v];
            if (slot < 0) {
                if ((int)s.ids.size() >= max_uniques) { s.overflow = true; return; }
                slot = (int)s.ids.size();
                s.ids.push_back((uint64_t)v); s.first.push_back(r);
            }
            code = slot;
        } else {
            size_t slot = 0;
            code = table.find((uint64_t)v, &slot);
            if (code < 0) {
                if ((int)s.ids.size() >= max_uniques) { s.overflow = true; return; }
                code = (int)s.ids.size();
                table.put(slot, (uint64_t)v, code);
                s.ids.push_back((uint64_t)v); s.first.push_back(r);
            }
        }
        codes[r] = code;
        last = v; last_code = code;
    }
}

template <typename T>
int label_codes(const T *ids, long long n, int max_uniques, int n_threads, int *codes, long long *first_rows, int *n_uniques) {
    int T_ = n_threads;
    if ((long long)T_ * 65536 > n) T_ = (int)(n / 65536);        // a slice below 64 K rows is not worth a thread
    if (T_ < 1) T_ = 1;
    std::vector<Slice> slices((size_t)T_);
    for (int t = 0; t < T_; ++t) { slices[(size_t)t].begin = n * t / T_; slices[(size_t)t].end = n * (t + 1) / T_; }
    std::vector<std::thread> pool;
    int started = 1;
    try {
        for (; started < T_; ++started) pool.emplace_back(encode_slice<T>, ids, std::ref(slices[(size_t)started]), max_uniques, codes);
    } catch (const std::system_error &) {}                       // no more threads to be had: the remaining slices run here
    encode_slice<T>(ids, slices[0], max_uniques, codes);
    for (int t = started; t < T_; ++t) encode_slice<T>(ids, slices[(size_t)t], max_uniques, codes);
    for (auto &th : pool) th.join();
    pool.clear();
    // merge in slice order: an id is numbered by the first slice that holds it, and inside a slice in its local order
    IdTable global;
    global.init();
    int next = 0;
    for (int t = 0; t < T_; ++t) {
        Slice &s = slices[(size_t)t];
        if (s.overflow) return abi_fail(PILOT_OT_ENOTSUP, "more than %d distinct labels", max_uniques);
        s.remap.resize(s.ids.size());
        for (size_t j = 0; j < s.ids.size(); ++j) {
            size_t slot = 0;
            int g = global.find(s.ids[j], &slot);
            if (g < 0) {
                if (next >= max_uniques) return abi_fail(PILOT_OT_ENOTSUP, "more than %d distinct labels", max_uniques);
                g = next++;
                global.put(slot, s.ids[j], g);
                first_rows[g] = s.first[j];
            }
            s.remap[j] = g;
        }
    }
    auto renumber = [&](int t) {
        const Slice &s = slices[(size_t)t];
        bool identity = true;
        for (size_t j = 0; j < s.remap.size(); ++j) identity = identity && s.remap[j] == (int)j;
        if (identity) return;
        const int *m = s.remap.data();
        for (long long r = s.begin; r < s.end; ++r) { const int c = codes[r]; if (c >= 0) codes[r] = m[c]; }
    };
    started = 2;
    try {
        for (; started < T_; ++started) pool.emplace_back(renumber, started);
    } catch (const std::system_error &) {}
    if (T_ > 1) renumber(1);
    for (int t = started; t < T_; ++t) renumber(t);
    for (auto &th : pool) th.join();
    *n_uniques = next;
    return PILOT_OT_OK;
}

}  // namespace

PILOT_API int pilot_ot_label_codes(const void *ids, int id_bytes, long long n, int max_uniques, int n_threads, int *codes,
                                   long long *first_rows, int *n_uniques) {
    if (!ids || !codes || !first_rows || !n_uniques) return abi_fail(PILOT_OT_EINVAL, "NULL pointer");
    if (n < 0 || max_uniques < 1) return abi_fail(PILOT_OT_EINVAL, "n=%lld max_uniques=%d out of range", n, max_uniques);
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 64) n_threads = 64;
    try {
        switch (id_bytes) {
            case 1: return label_codes(static_cast<const int8_t *>(ids), n, max_uniques, n_threads, codes, first_rows, n_uniques);
            case 2: return label_codes(static_cast<const int16_t *>(ids), n, max_uniques, n_threads, codes, first_rows, n_uniques);
            case 4: return label_codes(static_cast<const int32_t *>(ids), n, max_uniques, n_threads, codes, first_rows, n_uniques);
            case 8: return label_codes(static_cast<const uint64_t *>(ids), n, max_uniques, n_threads, codes, first_rows, n_uniques);
            default: return abi_fail(PILOT_OT_EINVAL, "id_bytes=%d: 1, 2, 4 (signed category codes) or 8 (opaque identities)", id_bytes);
        }
    } catch (const std::exception &e) {
        return abi_fail(PILOT_OT_EINVAL, "label_codes: %s", e.what());
    }
}
