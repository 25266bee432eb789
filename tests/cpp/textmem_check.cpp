// hgx_textmem (the memory of the texts the library hands out): blocks of a megabyte or more are mappings that grow in place or by
// moving with their contents, malloc's blocks are told apart, and released blocks are kept most-recent-first — two kept blocks, the
// older one making room — so that a text of the size released last finds its block again.  Built and run by tests/test_capi_host.py.
#include "hgx_textmem.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <thread>
#include <vector>
using namespace hgx;
#define CHECK(c)                                                   \
    do {                                                           \
        if (!(c)) {                                                \
            printf("FAILED line %d: %s\n", __LINE__, #c);          \
            return 1;                                              \
        }                                                          \
    } while (0)
int main() {
    const size_t MB = 1 << 20;
    char *s = (char *)textAlloc(1000); // malloc's
    CHECK(s && !textOwns(s));
    textFree(s);
    char *a = (char *)textAlloc(3 * MB);
    CHECK(a && textOwns(a));
    memset(a, 'a', 3 * MB);
    char *b = (char *)textRealloc(a, 40 * MB); // grows, contents kept
    CHECK(b && textOwns(b));
    for (size_t i = 0; i < 3 * MB; i += 4097)
        CHECK(b[i] == 'a');
    memset(b + 3 * MB, 'b', 37 * MB);
    CHECK((char *)textRealloc(b, 10 * MB) == b); // never shrinks
    textFree(b);
    char *c = (char *)textAlloc(30 * MB); // the block released last is found again (40 MB holds 30 without being four times too large)
    CHECK(c == b);
    textFree(c);
    // two small blocks kept, then a larger one released: the oldest goes, the larger one is kept and found again
    char *w = (char *)textAlloc(100 * MB), *m = (char *)textAlloc(60 * MB);
    CHECK(w && m);
    textFree(w);
    textFree(m);
    char *t = (char *)textAlloc(140 * MB);
    CHECK(t && t != w && t != m);
    t[0] = t[140 * MB - 1] = 'x';
    textFree(t);
    char *t2 = (char *)textAlloc(140 * MB);
    CHECK(t2 == t);
    textFree(t2);
    // growing from nothing is an allocation
    char *g = (char *)textRealloc(nullptr, 5 * MB);
    CHECK(g && textOwns(g));
    textFree(g);
    textTrim();
    char *f = (char *)textAlloc(5 * MB); // nothing kept any more: still fine
    CHECK(f && textOwns(f));
    textFree(f);
    textTrim();
    // several threads growing texts of their own at the same time (hgx_maf_export_multi's slices): a block that moves gives its old
    // addresses back, another thread's fresh mapping may land there at once, and each thread must still find its own block in the
    // registry — round 6: the mover erased the address AFTER the move, i.e. the newcomer's block, whose next growth went to realloc()
    {
        std::atomic<int> bad{0};
        auto worker = [&](int id) {
            for (int round = 0; round < 300 && !bad.load(); ++round) {
                size_t cap = MB;
                char *p = (char *)textRealloc(nullptr, cap);
                if (!p || !textOwns(p)) {
                    ++bad;
                    return;
                }
                p[0] = (char)id;
                for (int step = 0; step < 6; ++step) {
                    cap *= 2;
                    char *q = (char *)textRealloc(p, cap);
                    if (!q || !textOwns(q) || q[0] != (char)id) {
                        ++bad;
                        return;
                    }
                    p = q;
                    p[cap - 1] = (char)id;
                }
                if (!textOwns(p))
                    ++bad;
                textFree(p);
            }
        };
        std::vector<std::thread> ts;
        for (int id = 1; id <= 6; ++id)
            ts.emplace_back(worker, id);
        for (std::thread &t : ts)
            t.join();
        CHECK(bad.load() == 0);
        textTrim();
    }
    printf("same\n");
    return 0;
}
