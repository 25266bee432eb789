// CPU emulation of finish_wave's extractSegment (hal_amd/csrc/hgx_finish_kernel.hpp): the scalar loop the kernel had (every
// coordinate of every comparison read from the lanes) against the form on 64-bit masks it has now, member arrays in the place of
// lanes, on 2 M random sets whose target ranges are equal or disjoint (what the refinement leaves).  Prints "bad 0".
//   g++ -O2 -o /tmp/emul profiles/scripts/extract_segment_emulation.cpp && /tmp/emul
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <tuple>
typedef long long C;
enum { F_SREV = 1, F_TREV = 2 };
struct P { C tLo, tHi, sLo, sHi; int fl, seq; };
static int ffsll_(unsigned long long v) { return v ? __builtin_ctzll(v) + 1 : 0; }
typedef std::vector<std::pair<int,int>> Lines;
static Lines original(const std::vector<P> &p) {
    int n = (int)p.size();
    unsigned long long alive = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
    std::vector<C> cutv; Lines out;
    auto nextAlive = [&](int i) { if (i >= n) return n; unsigned long long mm = alive >> i; return mm ? i + ffsll_(mm) - 1 : n; };
    auto canMergeRight = [&](int a, int b) {
        if (((p[a].fl ^ p[b].fl) & 3) != 0) return false;
        if (p[b].tLo - p[a].tHi != 1) return false;
        bool same = ((p[a].fl & F_SREV) != 0) == ((p[a].fl & F_TREV) != 0);
        bool rOk = same ? (p[b].sLo - p[a].sHi == 1) : (p[a].sLo - p[b].sHi == 1);
        if (!rOk) return false;
        for (C c : cutv) if (c == p[a].tHi) return false;
        return true;
    };
    for (int i = nextAlive(0); i < n; i = nextAlive(i + 1)) {
        int v1s = i, v1n = 1, back = i, nxt = nextAlive(i + 1);
        while (nxt < n && p[back].tLo == p[nxt].tLo) { back = nxt; ++v1n; nxt = nextAlive(nxt + 1); }
        int fragBack = i; int seqI = p[i].seq;
        while (nxt < n) {
            int v2s = nxt, v2n = 0, b2 = -1;
            while (nxt < n && (v2n == 0 || p[b2].tLo == p[nxt].tLo) && v2n < v1n) { b2 = nxt; ++v2n; nxt = nextAlive(nxt + 1); }
            bool can = v1n == v2n; int a = v1s, b = v2s;
            for (int c = 0; c < v1n && can; ++c) { can = p[b].seq == seqI && canMergeRight(a, b); a = nextAlive(a + 1); b = nextAlive(b + 1); }
            if (!can) break;
            fragBack = v2s; alive &= ~(1ull << v2s); v1s = v2s; v1n = v2n;
        }
        if (v1n > 1) cutv.push_back(p[fragBack].tHi);
        out.push_back({i, fragBack});
    }
    return out;
}
static Lines masks(const std::vector<P> &p) {
    int n = (int)p.size();
    const unsigned long long all = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
    unsigned long long gb = 0;
    for (int l = 0; l < n; ++l) if (l == 0 || p[l].tLo != p[l-1].tLo) gb |= 1ull << l;
    std::vector<unsigned long long> mergeable(64, 0);
    for (int lane = 0; lane < n; ++lane) {
        const unsigned long long rest = lane < 63 ? (gb & all) >> (lane + 1) : 0ull;
        int nb = n, gsz = 0;
        if (rest) { int z = ffsll_(rest) - 1; nb = lane + 1 + z; unsigned long long rest2 = z < 63 ? rest >> (z + 1) : 0ull; gsz = (rest2 ? nb + ffsll_(rest2) : n) - nb; }
        bool same = ((p[lane].fl & F_SREV) != 0) == ((p[lane].fl & F_TREV) != 0);
        for (int c = 0; c < gsz; ++c) {
            int b = (nb + c) & 63;
            if (p[b].seq == p[lane].seq && ((p[lane].fl ^ p[b].fl) & 3) == 0 && p[b].tLo - p[lane].tHi == 1 && (same ? p[b].sLo - p[lane].sHi == 1 : p[lane].sLo - p[b].sHi == 1))
                mergeable[lane] |= 1ull << b;
        }
    }
    unsigned long long alive = all, cut = 0; Lines out;
    auto firstFrom = [&](unsigned long long m, int i) { if (i >= n) return n; unsigned long long mm = (m & all) >> i; return mm ? i + ffsll_(mm) - 1 : n; };
    auto span = [&](int lo, int hi) { unsigned long long upTo = hi >= 64 ? ~0ull : ((1ull << hi) - 1ull); return upTo & ~((1ull << lo) - 1ull); };
    for (int i = firstFrom(alive, 0); i < n; i = firstFrom(alive, i + 1)) {
        unsigned long long v1 = alive & span(i, firstFrom(gb, i + 1));
        int v1n = __builtin_popcountll(v1);
        int nxt = firstFrom(alive, firstFrom(gb, i + 1));
        int fragBack = i;
        while (nxt < n) {
            unsigned long long rest = alive & span(nxt, firstFrom(gb, nxt + 1)), v2 = 0; int v2n = 0, last = nxt;
            while (rest && v2n < v1n) { last = ffsll_(rest) - 1; v2 |= 1ull << last; rest &= rest - 1; ++v2n; }
            int after = firstFrom(alive, last + 1);
            bool can = v1n == v2n;
            for (unsigned long long m1 = v1, m2 = v2; m1 && can; m1 &= m1 - 1, m2 &= m2 - 1) {
                int a = ffsll_(m1) - 1, b = ffsll_(m2) - 1;
                can = ((mergeable[a] >> b) & 1) && !((cut >> a) & 1);
            }
            if (!can) break;
            fragBack = ffsll_(v2) - 1; alive &= ~(1ull << fragBack); v1 = v2; v1n = v2n; nxt = after;
        }
        if (v1n > 1) { int g0 = 63 - __builtin_clzll(gb & span(0, fragBack + 1)); cut |= span(g0, firstFrom(gb, fragBack + 1)); }
        out.push_back({i, fragBack});
    }
    return out;
}
int main() {
    srand(1); long bad = 0;
    for (int it = 0; it < 2000000; ++it) {
        // random classes: contiguous target ranges mostly adjacent; members with source coords
        int ncls = 1 + rand() % 8; std::vector<P> p; C t = 100;
        for (int k = 0; k < ncls && p.size() < 60; ++k) {
            if (rand() % 4 == 0) t += 1 + rand() % 3; // a hole
            C len = 1 + rand() % 5; int z = 1 + (rand() % 4 == 0 ? rand() % 3 : 0) + (rand() % 6 == 0);
            for (int m = 0; m < z; ++m) {
                P q; q.tLo = t; q.tHi = t + len - 1; q.fl = (rand() % 5 == 0 ? rand() % 4 : 0); q.seq = (t / 7) % 2;
                C s = (rand() % 3) * 1000 + (k * 5) + (rand() % 4 == 0 ? rand() % 7 : 0);
                // make some source-adjacent to the previous class's members: s continues
                q.sLo = s; q.sHi = s + len - 1; p.push_back(q);
            }
            t += len;
        }
        // make source coordinates chain for some: for each class k>0 member m, with prob set sLo = prev class member m's sHi+1
        // (recompute by scanning classes)
        std::vector<int> starts; for (int i = 0; i < (int)p.size(); ++i) if (i == 0 || p[i].tLo != p[i-1].tLo) starts.push_back(i);
        starts.push_back((int)p.size());
        for (size_t k = 1; k + 1 < starts.size(); ++k)
            for (int m = starts[k]; m < starts[k+1]; ++m) {
                int pm = starts[k-1] + (m - starts[k]);
                if (pm < starts[k] && rand() % 3) { C len = p[m].tHi - p[m].tLo + 1; bool same = ((p[pm].fl & 1) != 0) == ((p[pm].fl & 2) != 0);
                    if (same) { p[m].sLo = p[pm].sHi + 1; p[m].sHi = p[m].sLo + len - 1; } else { p[m].sHi = p[pm].sLo - 1; p[m].sLo = p[m].sHi - len + 1; }
                    if (rand() % 2) p[m].fl = p[pm].fl; }
            }
        std::sort(p.begin(), p.end(), [](const P &a, const P &b) { return std::tie(a.tLo, a.tHi, a.sLo, a.sHi) < std::tie(b.tLo, b.tHi, b.sLo, b.sHi); });
        if (original(p) != masks(p)) { if (++bad < 4) { printf("DIFF it=%d n=%zu\n", it, p.size()); for (auto &q : p) printf("  t[%lld,%lld] s[%lld,%lld] fl%d seq%d\n", q.tLo, q.tHi, q.sLo, q.sHi, q.fl, q.seq);
            for (auto &l : original(p)) printf(" o(%d,%d)", l.first, l.second); printf("\n"); for (auto &l : masks(p)) printf(" m(%d,%d)", l.first, l.second); printf("\n"); } }
    }
    printf("bad %ld\n", bad);
}
