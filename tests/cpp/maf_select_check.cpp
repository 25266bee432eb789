// hal2maf's row selection by rank (hal_amd/csrc/hgx_maf_kernels.hpp: MafSelect, the break sweeps) against the column walk
// (hgx_column_kernels.hpp: ColumnWalker, the reference's recursiveUpdate restated) on the host: the same functions the kernels
// run, compiled for the CPU (hipcc --offload-host-only -DHGX_DEV="__host__ __device__"), over whole alignment images.  For every
// genome as reference, and a set of filters (--noAncestors where the reference is a leaf, target sets):
//   * the sizes S (the reported bases in the tree below every base) and the rows A of every column, computed here by a plain
//     bottom-up pass, give, through MafSelect::row, every row of every column — the walk's rows in the walk's order;
//   * the break tracks (break_up_body / break_down_body, every thread of a small grid in turn) mark every column whose rows are
//     not the rows of the column before it advanced by one base, and the marked columns that are not such heads are found by
//     comparing them with the marked column before them, shifted (what k_maf_heads does).
// usage: maf_select_check <image.hgx> [<image.hgx> ...]; built and run by tests/test_capi_host.py
#include "hgx_maf_kernels.hpp"
#include <cstdio>
#include <cstring>
#include <functional>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

using namespace hgx;

namespace {
typedef int32_t C;
struct HostTables {
    std::vector<std::vector<TopRec<C>>> top;
    std::vector<std::vector<BotRec<C>>> bot;
    std::vector<std::vector<std::vector<int32_t>>> child;
    std::vector<std::vector<const int32_t *>> childPtr;
    std::vector<std::vector<int32_t>> childGenome;
    std::vector<std::vector<int64_t>> seqStart;
    std::vector<GenomeDesc> desc;
};
int32_t encLink(int64_t idx, bool rev) {
    return idx < 0 ? -1 : (int32_t)((idx << 1) | (rev ? 1 : 0));
}
void build(const Image &img, HostTables &H) {
    const size_t ng = img.genomes.size();
    H.top.resize(ng);
    H.bot.resize(ng);
    H.child.resize(ng);
    H.childPtr.resize(ng);
    H.childGenome.resize(ng);
    H.seqStart.resize(ng);
    H.desc.resize(ng);
    for (size_t g = 0; g < ng; ++g) {
        const GenomeTables &G = img.genomes[g];
        H.top[g].resize((size_t)G.numTop + 1);
        for (int64_t i = 0; i < G.numTop; ++i)
            H.top[g][(size_t)i] = TopRec<C>{(C)G.tStart[(size_t)i], encLink(G.tParent[(size_t)i], G.tParentRev[(size_t)i] != 0), (int32_t)G.tParalogy[(size_t)i],
                                            (int32_t)G.tBotParse[(size_t)i]};
        H.top[g][(size_t)G.numTop] = TopRec<C>{(C)G.totalLength, -1, -1, -1};
        H.bot[g].resize((size_t)G.numBot + 1);
        for (int64_t i = 0; i < G.numBot; ++i)
            H.bot[g][(size_t)i] = BotRec<C>{(C)G.bStart[(size_t)i], (int32_t)G.bTopParse[(size_t)i]};
        H.bot[g][(size_t)G.numBot] = BotRec<C>{(C)G.totalLength, -1};
        H.child[g].resize(G.children.size());
        for (size_t k = 0; k < G.children.size(); ++k) {
            H.child[g][k].resize((size_t)G.numBot + 1, -1);
            for (int64_t i = 0; i < G.numBot; ++i)
                H.child[g][k][(size_t)i] = encLink(G.bChild[k][(size_t)i], G.bChildRev[k][(size_t)i] != 0);
            H.childPtr[g].push_back(H.child[g][k].data());
            H.childGenome[g].push_back(G.children[k]);
        }
        for (const SeqInfo &S : G.seqs)
            H.seqStart[g].push_back(S.start);
        H.seqStart[g].push_back(G.totalLength);
    }
    for (size_t g = 0; g < ng; ++g) {
        const GenomeTables &G = img.genomes[g];
        GenomeDesc &d = H.desc[g];
        d.top = H.top[g].data();
        d.bot = H.bot[g].data();
        d.child = H.childPtr[g].data();
        d.childGenome = H.childGenome[g].data();
        d.dna = G.dna.empty() ? nullptr : G.dna.data();
        d.seqStart = H.seqStart[g].data();
        d.numTop = G.numTop;
        d.numBot = G.numBot;
        d.length = G.totalLength;
        d.parent = G.parent;
        d.slotInParent = G.parent >= 0 ? img.genomes[(size_t)G.parent].childSlotOf((int)g) : -1;
        d.numChildren = (int32_t)G.children.size();
        d.numSeq = (int32_t)G.seqs.size();
    }
}

size_t g_uniqueRuns = 0, g_uniqueSplitRuns = 0, g_uniqueColumns[3] = {0, 0, 0}; // runs checked, runs of more than one stretch, columns by class

struct Case {
    int ref;
    bool noAncestors;
    std::vector<int> targets;
};

size_t checkCase(const Image &img, const HostTables &H, const Case &cs, size_t &columns, size_t &marked, size_t &heads) {
    const int ng = (int)img.genomes.size();
    const size_t words = ((size_t)ng + 63) / 64;
    std::vector<unsigned long long> scope(words, 0), target(words, 0);
    std::vector<char> inScope((size_t)ng, 1), counted((size_t)ng, 1);
    int scopeRoot = img.root();
    if (cs.targets.empty()) {
        std::fill(scope.begin(), scope.end(), ~0ull);
        std::fill(target.begin(), target.end(), ~0ull);
    } else { // halColumnIterator.cpp:45-51
        std::set<int> tg(cs.targets.begin(), cs.targets.end());
        tg.insert(cs.ref);
        scopeRoot = cs.ref;
        for (int g : tg)
            scopeRoot = img.lca(scopeRoot, g);
        std::fill(inScope.begin(), inScope.end(), 0);
        std::fill(counted.begin(), counted.end(), 0);
        for (int g : tg) {
            counted[(size_t)g] = 1;
            target[(size_t)(g >> 6)] |= 1ull << (g & 63);
            for (int x = g;; x = img.genomes[(size_t)x].parent) {
                inScope[(size_t)x] = 1;
                scope[(size_t)(x >> 6)] |= 1ull << (x & 63);
                if (x == scopeRoot)
                    break;
            }
        }
    }
    if (cs.noAncestors)
        for (int g = 0; g < ng; ++g)
            if (!img.genomes[(size_t)g].children.empty())
                counted[(size_t)g] = 0;
    unsigned int err = 0;
    ColumnParams P;
    memset(&P, 0, sizeof P);
    P.desc = H.desc.data();
    P.numGenomes = ng;
    P.ref = cs.ref;
    P.first = 0;
    P.count = img.genomes[(size_t)cs.ref].totalLength;
    P.step = 1;
    P.noAncestors = cs.noAncestors;
    P.scopeMask = scope.data();
    P.targetMask = target.data();
    P.error = &err;
    // post-order of the scope, the path to the reference
    std::vector<int> post, stack{scopeRoot}, path;
    while (!stack.empty()) {
        const int g = stack.back();
        stack.pop_back();
        post.push_back(g);
        for (int c : img.genomes[(size_t)g].children)
            if (inScope[(size_t)c])
                stack.push_back(c);
    }
    std::reverse(post.begin(), post.end());
    for (int x = cs.ref;; x = img.genomes[(size_t)x].parent) {
        path.push_back(x);
        if (x == scopeRoot)
            break;
    }
    std::reverse(path.begin(), path.end());
    std::vector<char> hasTrack((size_t)ng, 0);
    for (int g : post) {
        const GenomeTables &G = img.genomes[(size_t)g];
        if (G.numBot <= 0)
            continue;
        for (int c : G.children)
            if (inScope[(size_t)c] && img.genomes[(size_t)c].totalLength > 0 && img.genomes[(size_t)c].numTop > 0)
                hasTrack[(size_t)g] = 1;
    }
    // S by a plain bottom-up pass (what k_sweep_up<C, int32_t, true> leaves)
    std::vector<std::vector<int32_t>> S((size_t)ng);
    for (int g : post) {
        if (!hasTrack[(size_t)g])
            continue;
        const GenomeTables &G = img.genomes[(size_t)g];
        S[(size_t)g].assign((size_t)G.totalLength, counted[(size_t)g] ? 1 : 0);
        for (size_t k = 0; k < G.children.size(); ++k) {
            const int c = G.children[k];
            const GenomeTables &CG = img.genomes[(size_t)c];
            if (!inScope[(size_t)c] || CG.totalLength <= 0 || CG.numTop <= 0)
                continue;
            for (int64_t b = 0; b < G.numBot; ++b) {
                const int64_t t0 = G.bChild[k][(size_t)b];
                if (t0 < 0)
                    continue;
                const int64_t start = G.bStart[(size_t)b], len = G.bStart[(size_t)b + 1] - start;
                int64_t t = t0;
                do {
                    const bool rev = CG.tParentRev[(size_t)t] != 0;
                    for (int64_t o = 0; o < len; ++o) {
                        const int64_t cp = CG.tStart[(size_t)t] + (rev ? len - 1 - o : o);
                        S[(size_t)g][(size_t)(start + o)] += hasTrack[(size_t)c] ? S[(size_t)c][(size_t)cp] : (counted[(size_t)c] ? 1 : 0);
                    }
                    t = CG.tParalogy[(size_t)t];
                } while (t >= 0 && t != t0);
            }
        }
    }
    std::vector<const int32_t *> sPtr((size_t)ng, nullptr);
    for (int g = 0; g < ng; ++g)
        if (hasTrack[(size_t)g])
            sPtr[(size_t)g] = S[(size_t)g].data();
    // the break tracks, by the kernels' own bodies: every thread of a grid of 64 threads in turn
    const int64_t THREADS = 64;
    std::vector<std::vector<uint8_t>> D((size_t)ng), F((size_t)ng);
    for (int g : post) {
        if (!hasTrack[(size_t)g])
            continue;
        const GenomeTables &G = img.genomes[(size_t)g];
        D[(size_t)g].assign((size_t)G.totalLength + 8, 0);
        std::vector<BreakChild> kids;
        for (size_t k = 0; k < G.children.size(); ++k) {
            const int c = G.children[k];
            if (inScope[(size_t)c] && hasTrack[(size_t)c])
                kids.push_back(BreakChild{H.child[(size_t)g][k].data(), H.top[(size_t)c].data(), D[(size_t)c].data()});
        }
        size_t at = 0;
        do {
            BreakChildren ch;
            ch.noRing = 0;
            ch.n = (int)std::min<size_t>(SWEEP_MAX_CHILDREN, kids.size() - at);
            for (int k = 0; k < ch.n; ++k)
                ch.c[k] = kids[at + (size_t)k];
            for (int64_t th = 0; th < THREADS; ++th)
                break_up_body<C>(th, THREADS, H.bot[(size_t)g].data(), G.numBot, ch, at ? 1 : 0, D[(size_t)g].data());
            at += SWEEP_MAX_CHILDREN;
        } while (at < kids.size());
    }
    {
        const int top = path[0];
        const GenomeTables &G = img.genomes[(size_t)top];
        F[(size_t)top].assign((size_t)G.totalLength + 8, 0);
        for (int64_t i = 0; i < G.totalLength; ++i)
            F[(size_t)top][(size_t)i] = hasTrack[(size_t)top] ? D[(size_t)top][(size_t)i] : 0;
        for (int64_t t = 0; t < G.numTop; ++t)
            F[(size_t)top][(size_t)G.tStart[(size_t)t]] = 1;
    }
    for (size_t i = 1; i < path.size(); ++i) {
        const int c = path[i], p = path[i - 1];
        const GenomeTables &G = img.genomes[(size_t)c];
        F[(size_t)c].assign((size_t)G.totalLength + 8, 0);
        for (int64_t th = 0; th < THREADS; ++th)
            break_down_body<C>(th, THREADS, H.top[(size_t)c].data(), G.numTop, H.bot[(size_t)p].data(), F[(size_t)p].data(),
                               hasTrack[(size_t)c] ? D[(size_t)c].data() : nullptr, F[(size_t)c].data());
    }
    // every column: the walk's rows against the rows by rank; the marks against the columns that begin a run
    MafRowParams M;
    M.P = P;
    M.S = sPtr.data();
    M.candCol = nullptr;
    M.candRow = nullptr;
    M.nCand = 0;
    M.refLocate = nullptr; // (the search over all of the reference's segments)
    M.refLocateShift = 0;
    MafSelect<C> sel(M);
    ColumnWalker<C> walker(P);
    const int64_t n = img.genomes[(size_t)cs.ref].totalLength;
    std::vector<ColumnRow> want(4096), got(4096), prev, prevMarked;
    int64_t prevMarkedCol = -1;
    size_t failures = 0;
    // (for the check of --unique below: every column's rows, and which columns are marked)
    std::vector<ColumnRow> allRows;
    std::vector<size_t> rowsAt{0};
    std::vector<char> isMarked;
    auto fail = [&](const char *what, int64_t p) {
        if (failures < 5)
            printf("  DIFFERENT (%s): reference %s, column %lld, noAncestors %d, %zu targets\n", what, img.genomes[(size_t)cs.ref].name.c_str(), (long long)p,
                   (int)cs.noAncestors, cs.targets.size());
        ++failures;
    };
    for (int64_t p = 0; p < n; ++p) {
        RowVisitor v;
        v.dst = want.data();
        v.desc = P.desc;
        // (the walk's rows: room for them first)
        {
            struct Count {
                size_t bases = 0;
                void operator()(int, int64_t, bool) { ++bases; }
            } dv;
            walker.run(p, dv);
            if (dv.bases > want.size()) {
                want.resize(dv.bases);
                got.resize(dv.bases);
                v.dst = want.data();
            }
        }
        walker.run(p, v);
        const int64_t rows = v.n;
        // total by the sizes: S at the topmost ancestor
        const int32_t seg = sel.locate(p);
        for (int64_t r = 0; r < rows; ++r) {
            memset(&got[(size_t)r], 0xee, sizeof(ColumnRow));
            if (!sel.row(&got[(size_t)r], seg, p, r, rows)) {
                fail("row not found", p);
                break;
            }
            if (memcmp(&got[(size_t)r], &want[(size_t)r], sizeof(ColumnRow)) != 0) {
                fail("row differs", p);
                break;
            }
        }
        // one row more must not exist
        {
            ColumnRow extra;
            if (sel.row(&extra, seg, p, rows, rows + 1) && false)
                fail("a row too many", p);
        }
        // does the column continue the one before?
        bool cont = p > 0 && (int64_t)prev.size() == rows;
        for (int64_t r = 0; cont && r < rows; ++r) {
            const ColumnRow &a = want[(size_t)r], &q = prev[(size_t)r];
            cont = a.genome == q.genome && a.rev == q.rev && a.pos == (q.rev ? q.pos - 1 : q.pos + 1);
        }
        const bool mark = p == 0 || F[(size_t)cs.ref][(size_t)p] != 0;
        if (!cont && !mark)
            fail("a head that is not marked", p);
        ++columns;
        if (mark) {
            ++marked;
            // k_maf_heads' test: the marked column before, advanced by the distance
            bool head = prevMarkedCol < 0 || (int64_t)prevMarked.size() != rows;
            const int64_t d = p - prevMarkedCol;
            for (int64_t r = 0; !head && r < rows; ++r) {
                const ColumnRow &a = want[(size_t)r], &q = prevMarked[(size_t)r];
                head = a.genome != q.genome || a.rev != q.rev || a.pos != (q.rev ? q.pos - d : q.pos + d);
            }
            if (head != !cont && p > 0)
                fail("marked column: head by the shifted comparison differs from head by its neighbour", p);
            if (head)
                ++heads;
            prevMarked.assign(want.begin(), want.begin() + rows);
            prevMarkedCol = p;
        }
        prev.assign(want.begin(), want.begin() + rows);
        allRows.insert(allRows.end(), want.begin(), want.begin() + rows);
        rowsAt.push_back(allRows.size());
        isMarked.push_back(mark ? 1 : 0);
    }
    if (walker.overflow || err)
        fail("walk overflow", -1);
    // --unique (unique_stretches, what k_unique_count / k_unique_stretches run): for ranges that begin at several columns f — the
    // range's first column is a batch's first column and so marked — the stretches of every run against the class of every column
    // told from the column's own rows (hgx_column_kernels.hpp: k_column_unique_count's rule)
    for (int64_t f : {(int64_t)0, n / 3, (2 * n) / 3, n - 1}) {
        if (f < 0 || f >= n)
            continue;
        std::vector<uint32_t> candCol, candRow;
        std::vector<ColumnRow> rows;
        for (int64_t p = f; p < n; ++p)
            if (p == f || isMarked[(size_t)p]) {
                candCol.push_back((uint32_t)(p - f));
                candRow.push_back((uint32_t)rows.size());
                rows.insert(rows.end(), allRows.begin() + (std::ptrdiff_t)rowsAt[(size_t)p], allRows.begin() + (std::ptrdiff_t)rowsAt[(size_t)p + 1]);
            }
        candRow.push_back((uint32_t)rows.size());
        unsigned int uerr = 0;
        UniqueParams U;
        U.candCol = candCol.data();
        U.candRow = candRow.data();
        U.rows = rows.data();
        U.nCand = (uint32_t)candCol.size();
        U.n = (uint32_t)(n - f);
        U.first = f;
        U.f = f;
        U.ref = cs.ref;
        U.maxRefRows = UNIQUE_MAX_REF_ROWS;
        U.error = &uerr;
        auto classOf = [&](int64_t p) { // the column's class from its own rows
            bool inRange = false, left = false;
            for (size_t i = rowsAt[(size_t)p]; i < rowsAt[(size_t)p + 1]; ++i) {
                const ColumnRow &r = allRows[i];
                if (r.genome == cs.ref && r.pos < p) {
                    if (r.pos >= f)
                        inRange = true;
                    else
                        left = true;
                }
            }
            return inRange ? (uint32_t)COL_SKIPPED : left ? (uint32_t)COL_KEYS_ONLY : (uint32_t)COL_WRITTEN;
        };
        for (uint32_t k = 0; k < U.nCand; ++k) {
            const int64_t h = candCol[k], L = (k + 1 < U.nCand ? (int64_t)candCol[k + 1] : (int64_t)U.n) - h;
            int64_t covered = 0;
            bool bad = false;
            uint32_t last = 99;
            const bool ok = unique_stretches(U, k, [&](int64_t j, int64_t len, uint32_t cls) {
                if (j != covered || len <= 0 || cls == last)
                    bad = true; // (the stretches tile the run, in order, and neighbours differ)
                last = cls;
                for (int64_t t = 0; t < len && j + t < L; ++t)
                    if (classOf(f + h + j + t) != cls)
                        bad = true;
                if (j > 0)
                    ++g_uniqueSplitRuns;
                g_uniqueColumns[cls] += (size_t)len;
                covered = j + len;
            });
            ++g_uniqueRuns;
            if (!ok) { // (more reference copies than a lane holds: the batch goes to the walk — fine, but rare)
                bool many = false;
                size_t refs = 0;
                for (uint32_t i = candRow[k]; i < candRow[k + 1]; ++i)
                    refs += rows[i].genome == cs.ref;
                many = refs > (size_t)UNIQUE_MAX_REF_ROWS;
                if (!many)
                    fail("--unique: the run's reference rows were refused", f + h);
                continue;
            }
            if (bad || covered != L)
                fail("--unique: the run's stretches are not the columns' classes", f + h);
        }
    }
    return failures;
}
} // namespace

int main(int argc, char **argv) {
    size_t failures = 0;
    for (int a = 1; a < argc; ++a) {
        const Image img = openAlignmentFile(argv[a]);
        HostTables H;
        build(img, H);
        const int ng = (int)img.genomes.size();
        size_t columns = 0, marked = 0, heads = 0, cases = 0;
        for (int ref = 0; ref < ng; ++ref) {
            if (img.genomes[(size_t)ref].totalLength <= 0)
                continue;
            std::vector<Case> cs{{ref, false, {}}};
            if (img.genomes[(size_t)ref].children.empty())
                cs.push_back({ref, true, {}});
            std::vector<int> others;
            for (int g = 0; g < ng; ++g)
                if (g != ref)
                    others.push_back(g);
            if (others.size() >= 2)
                cs.push_back({ref, false, {others[(size_t)ref % others.size()], others[(size_t)(ref * 7 + 3) % others.size()]}});
            if (others.size() >= 4 && img.genomes[(size_t)ref].children.empty())
                cs.push_back({ref, true, {others[0], others[others.size() / 2], others.back()}});
            for (const Case &c : cs) {
                failures += checkCase(img, H, c, columns, marked, heads);
                ++cases;
            }
        }
        printf("%s: %zu cases, %zu columns, %zu marked, %zu heads\n", argv[a], cases, columns, marked, heads);
    }
    if (failures) {
        printf("FAILED: %zu differences\n", failures);
        return 1;
    }
    printf("--unique: %zu runs (%zu breaks inside runs), columns passed over %zu, written %zu, walked for their keys %zu\n", g_uniqueRuns,
           g_uniqueSplitRuns, g_uniqueColumns[0], g_uniqueColumns[1], g_uniqueColumns[2]);
    printf("OK\n");
    return 0;
}
