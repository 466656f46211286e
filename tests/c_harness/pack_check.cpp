// pack_check.cpp — CPU check of csrc/chain_pack.h (the per-batch job packing of graph_chain2.hip's device kernel): the SAME functions the
// kernel calls, driven sequentially (thread loops become for loops), against a brute-force best fit decreasing.
//   * every member graph gets exactly one (job, first slot); inside a job the graphs tile slots 0 .. used - 1 without a gap or an overlap
//     and never pass slot 63;
//   * the number of jobs equals sequential best fit decreasing's (what gnnmp_chain_jobs_create's host loop produces);
//   * the record count stays inside PACK_MAX_REC.
// Built and run by tests/test_chain_pack_cpu.py (g++, no GPU, no HIP).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "chain_pack.h"

using namespace gnnmp;

static int brute_bfd_jobs(const std::vector<int> &sizes) {
    std::vector<int> order(sizes.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sizes[a] > sizes[b]; });
    std::vector<int> room;
    for (int g : order) {
        const int s = sizes[g];
        int best = -1;
        for (size_t j = 0; j < room.size(); ++j)
            if (room[j] >= s && (best < 0 || room[j] < room[best])) best = (int)j;
        if (best < 0) { room.push_back(64); best = (int)room.size() - 1; }
        room[best] -= s;
    }
    return (int)room.size();
}

static int check(const std::vector<int> &sizes, const char *tag) {
    const int G = (int)sizes.size();
    std::vector<int32_t> cnt(66, 0), start(66, 0);
    for (int s : sizes) ++cnt[s];
    {   // sizes descending
        int pos = 0;
        for (int s = 64; s >= 1; --s) { start[s] = pos; pos += cnt[s]; }
    }
    // the size-sorted order (stable by id), as the kernel's counting sort produces it
    std::vector<int32_t> sorted(G), fill(start.begin(), start.end());
    for (int g = 0; g < G; ++g) sorted[fill[sizes[g]]++] = g;
    static PackState st;
    pack_records(cnt.data(), start.data(), &st);
    if (st.nrec > PACK_MAX_REC) { printf("%s: %d records\n", tag, st.nrec); return 1; }
    int32_t counts[65];
    for (int d = 0; d <= 64; ++d) pack_class_count(&st, d, &counts[d]);
    st.cls_start[0] = 0;
    for (int d = 0; d <= 64; ++d) st.cls_start[d + 1] = st.cls_start[d] + counts[d];
    for (int d = 0; d <= 64; ++d) pack_class_fill(&st, d);
    std::vector<std::vector<std::pair<int, int>>> jobs((size_t)st.njobs);    // (slot, size)
    for (int p = 0; p < G; ++p) {
        int32_t job, slot;
        pack_place(&st, p, &job, &slot);
        if (job < 0 || job >= st.njobs) { printf("%s: graph at %d -> job %d of %d\n", tag, p, job, st.njobs); return 1; }
        jobs[(size_t)job].push_back({slot, sizes[sorted[p]]});
    }
    int tiles = 0;
    for (int j = 0; j < st.njobs; ++j) {
        auto &v = jobs[(size_t)j];
        std::sort(v.begin(), v.end());
        int at = 0;
        for (auto &pr : v) {
            if (pr.first != at) { printf("%s: job %d slot %d expected %d\n", tag, j, pr.first, at); return 1; }
            at += pr.second;
        }
        if (at > 64 || at == 0) { printf("%s: job %d holds %d rows\n", tag, j, at); return 1; }
        tiles += at > 32 ? 2 : 1;
    }
    if (tiles != st.tiles) { printf("%s: tiles %d vs %d\n", tag, tiles, st.tiles); return 1; }
    const int want = brute_bfd_jobs(sizes);
    if (want != st.njobs) { printf("%s: %d jobs, best fit decreasing has %d\n", tag, st.njobs, want); return 1; }
    printf("%-28s G %6d  jobs %5d  records %4d  tiles %5d  ok\n", tag, G, st.njobs, st.nrec, st.tiles);
    return 0;
}

int main() {
    std::mt19937 rng(12345);
    int bad = 0;
    auto uni = [&](int lo, int hi, int G) {
        std::vector<int> v((size_t)G);
        for (int &x : v) x = lo + (int)(rng() % (unsigned)(hi - lo + 1));
        return v;
    };
    bad += check(uni(20, 40, 8192), "U{20..40} x 8192");
    bad += check(uni(1, 64, 8192), "U{1..64} x 8192");
    bad += check(uni(1, 64, 3000), "U{1..64} x 3000");
    bad += check(uni(1, 8, 5000), "U{1..8} x 5000");
    bad += check(uni(33, 64, 2000), "U{33..64} x 2000");
    bad += check(uni(30, 34, 4097), "U{30..34} x 4097");
    bad += check(std::vector<int>(1000, 64), "all 64");
    bad += check(std::vector<int>(1000, 1), "all 1");
    bad += check(std::vector<int>(777, 21), "all 21");
    bad += check(std::vector<int>(1, 5), "one graph");
    {
        std::vector<int> v;                      // every size, a prime number of times each
        for (int s = 1; s <= 64; ++s)
            for (int i = 0; i < 1 + (s * 7) % 13; ++i) v.push_back(s);
        std::shuffle(v.begin(), v.end(), rng);
        bad += check(v, "every size");
    }
    for (int trial = 0; trial < 200; ++trial) {           // small random batches, skewed size laws
        const int G = 1 + (int)(rng() % 300);
        std::vector<int> v((size_t)G);
        const int mode = trial % 4;
        for (int &x : v) {
            const unsigned r = rng();
            x = mode == 0 ? 1 + (int)(r % 64) : mode == 1 ? 1 + (int)((r % 64) * (r % 64) / 64) : mode == 2 ? 64 - (int)((r % 64) * (r % 64) / 64) * 63 / 63
                                                                                                                 : 1 + (int)(r % 3) * 20 + (int)((r >> 8) % 4);
            if (x < 1) x = 1;
            if (x > 64) x = 64;
        }
        char tag[64];
        snprintf(tag, sizeof(tag), "random %d (mode %d)", trial, mode);
        const int b = check(v, tag);
        bad += b;
        if (b) break;
    }
    if (bad) { printf("FAILED\n"); return 1; }
    printf("all packings valid and as tight as best fit decreasing\n");
    return 0;
}
