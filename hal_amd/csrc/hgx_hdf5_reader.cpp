// Read-only importer for HDF5-format HAL files (the format most production HAL files use): reads the datasets of
// /root/reference/api/hdf5_impl whole and transposes them into the flat image (hgx_image.hpp).
//   file layout   hdf5Alignment.cpp:36-39: groups "Meta", "Phylogeny" (attribute "Phylogeny" = Newick text, variable-length
//                 string, hdf5MetaData.cpp:56-64, hdf5Alignment.cpp:622-626), "Verison" [sic], an unused "Genomes", and one group per genome at the file root
//   genome group  hdf5Genome.cpp:29-35: datasets DNA_ARRAY (uint8, two bases per byte, hdf5Genome.h:39-41),
//                 TOP_ARRAY, BOTTOM_ARRAY, SEQIDX_ARRAY, SEQNAME_ARRAY
//   top segment   hdf5TopSegment.cpp:19-24,71-87: packed 33-byte records {genomeIdx (start), bottomIdx (bottom parse),
//                 paralogyIdx, parentIdx: int64; reverseFlag: char}, numTop+1 of them (hdf5Genome.cpp:292)
//   bottom seg.   hdf5BottomSegment.cpp:18-31,64-94: {genomeIdx (start) int64, length u64, topIdx int64,
//                 numChildren x {childIdx int64, reverseFlag char}}, numBottom+1 of them (hdf5Genome.cpp:319)
//   sequences     hdf5Sequence.cpp:25-53: SEQIDX_ARRAY {start, topSegmentArrayIndexOffset, bottomSegmentArrayIndexOffset:
//                 u64} x (numSequences+1), lengths and counts are differences to the next record (:96-115);
//                 SEQNAME_ARRAY fixed-length strings
// Child slot k of a genome = its k-th child in the Newick text (halGenome.h:247-256), as in the mmap format.
//
// libhdf5 is loaded at run time (dlopen) so that libhgx.so carries no link dependency on it; only its C API is used.
// The handful of prototypes below follow the HDF5 >= 1.10 ABI (hid_t is a 64-bit integer); the library version is
// checked after loading.  Records are read with the file's own datatype as memory type, i.e. as the packed
// little-endian bytes the reference wrote.
#include "hgx_image.hpp"
#include <cstring>
#include <deque>
#include <dlfcn.h>
#include <map>
#include <mutex>

namespace hgx {

namespace {

typedef int64_t hid_t;
typedef int herr_t;
typedef int htri_t;
typedef long long hssize_t;

struct H5 {
    void *lib = nullptr;
    herr_t (*open)() = nullptr;
    herr_t (*get_libversion)(unsigned *, unsigned *, unsigned *) = nullptr;
    herr_t (*Eset_auto2)(hid_t, void *, void *) = nullptr;
    hid_t (*Fopen)(const char *, unsigned, hid_t) = nullptr;
    herr_t (*Fclose)(hid_t) = nullptr;
    hid_t (*Gopen2)(hid_t, const char *, hid_t) = nullptr;
    herr_t (*Gclose)(hid_t) = nullptr;
    htri_t (*Lexists)(hid_t, const char *, hid_t) = nullptr;
    htri_t (*Aexists)(hid_t, const char *) = nullptr;
    hid_t (*Aopen)(hid_t, const char *, hid_t) = nullptr;
    herr_t (*Aread)(hid_t, hid_t, void *) = nullptr;
    herr_t (*Aclose)(hid_t) = nullptr;
    hid_t (*Tcopy)(hid_t) = nullptr;
    herr_t (*Tset_size)(hid_t, size_t) = nullptr;
    size_t (*Tget_size)(hid_t) = nullptr;
    herr_t (*Tclose)(hid_t) = nullptr;
    hid_t (*Dopen2)(hid_t, const char *, hid_t) = nullptr;
    hid_t (*Dget_type)(hid_t) = nullptr;
    hid_t (*Dget_space)(hid_t) = nullptr;
    hssize_t (*Sget_simple_extent_npoints)(hid_t) = nullptr;
    herr_t (*Sclose)(hid_t) = nullptr;
    herr_t (*Dread)(hid_t, hid_t, hid_t, hid_t, hid_t, void *) = nullptr;
    herr_t (*Dclose)(hid_t) = nullptr;
    herr_t (*free_memory)(void *) = nullptr;
    hid_t C_S1 = -1; // H5T_C_S1

    template <typename F> void sym(F &f, const char *name) {
        f = (F)dlsym(lib, name);
        if (!f)
            throw std::runtime_error(std::string("HDF5 library lacks symbol ") + name);
    }
    void load() {
        const char *env = getenv("HGX_HDF5_LIB");
        const char *candidates[] = {env, "libhdf5.so.103", "libhdf5.so", "/opt/conda/lib/libhdf5.so.103", "/opt/conda/lib/libhdf5.so",
                                    "libhdf5_serial.so.103", "libhdf5_serial.so"};
        std::string tried;
        for (const char *c : candidates) {
            if (!c || !*c)
                continue;
            lib = dlopen(c, RTLD_NOW | RTLD_LOCAL);
            if (lib)
                break;
            tried += std::string(" ") + c;
        }
        if (!lib)
            throw std::runtime_error("HDF5-format HAL needs the HDF5 C library at run time; none of" + tried +
                                     " could be loaded (set HGX_HDF5_LIB, or convert the file with `halExtract --outputFormat mmap`)");
        sym(open, "H5open");
        sym(get_libversion, "H5get_libversion");
        sym(Eset_auto2, "H5Eset_auto2");
        sym(Fopen, "H5Fopen");
        sym(Fclose, "H5Fclose");
        sym(Gopen2, "H5Gopen2");
        sym(Gclose, "H5Gclose");
        sym(Lexists, "H5Lexists");
        sym(Aexists, "H5Aexists");
        sym(Aopen, "H5Aopen");
        sym(Aread, "H5Aread");
        sym(Aclose, "H5Aclose");
        sym(Tcopy, "H5Tcopy");
        sym(Tset_size, "H5Tset_size");
        sym(Tget_size, "H5Tget_size");
        sym(Tclose, "H5Tclose");
        sym(Dopen2, "H5Dopen2");
        sym(Dget_type, "H5Dget_type");
        sym(Dget_space, "H5Dget_space");
        sym(Sget_simple_extent_npoints, "H5Sget_simple_extent_npoints");
        sym(Sclose, "H5Sclose");
        sym(Dread, "H5Dread");
        sym(Dclose, "H5Dclose");
        sym(free_memory, "H5free_memory");
        unsigned maj = 0, min = 0, rel = 0;
        if (open() < 0 || get_libversion(&maj, &min, &rel) < 0)
            throw std::runtime_error("HDF5 library failed to initialise");
        if (maj != 1 || min < 10)
            throw std::runtime_error("HDF5 library " + std::to_string(maj) + "." + std::to_string(min) +
                                     " is too old: the reader is written against the 1.10+ ABI");
        hid_t *s1 = (hid_t *)dlsym(lib, "H5T_C_S1_g");
        if (!s1)
            throw std::runtime_error("HDF5 library lacks symbol H5T_C_S1_g");
        C_S1 = *s1;
        Eset_auto2(0 /* H5E_DEFAULT */, nullptr, nullptr); // errors are reported through exceptions, not HDF5's stderr dump
    }
};

H5 &h5() {
    static H5 lib;
    static std::once_flag once;
    static std::string failure;
    std::call_once(once, [&]() {
        try {
            lib.load();
        } catch (const std::exception &e) {
            failure = e.what();
        }
    });
    if (!failure.empty())
        throw std::runtime_error(failure);
    return lib;
}

struct Handle { // closes with the matching H5?close
    hid_t id = -1;
    herr_t (*closer)(hid_t) = nullptr;
    Handle(hid_t i, herr_t (*c)(hid_t)) : id(i), closer(c) {}
    Handle(const Handle &) = delete;
    ~Handle() {
        if (id >= 0 && closer)
            closer(id);
    }
};

// variable-length string attribute `name` of group `group` (HDF5MetaData, hdf5MetaData.cpp:56-64)
std::string readStringAttr(H5 &L, hid_t group, const std::string &name, const std::string &what) {
    if (L.Aexists(group, name.c_str()) <= 0)
        throw std::runtime_error(what + ": attribute '" + name + "' not found");
    Handle attr(L.Aopen(group, name.c_str(), 0), L.Aclose);
    Handle type(L.Tcopy(L.C_S1), L.Tclose);
    if (attr.id < 0 || type.id < 0 || L.Tset_size(type.id, (size_t)-1 /* H5T_VARIABLE */) < 0)
        throw std::runtime_error(what + ": cannot open attribute '" + name + "'");
    char *s = nullptr;
    if (L.Aread(attr.id, type.id, &s) < 0 || !s)
        throw std::runtime_error(what + ": cannot read attribute '" + name + "'");
    std::string out(s);
    L.free_memory(s);
    return out;
}

// whole dataset as raw records in the file's datatype; returns the record size
size_t readDataset(H5 &L, hid_t group, const char *name, std::vector<uint8_t> &bytes, size_t &count, const std::string &what) {
    Handle d(L.Dopen2(group, name, 0), L.Dclose);
    if (d.id < 0)
        throw std::runtime_error(what + ": cannot open dataset " + name);
    Handle t(L.Dget_type(d.id), L.Tclose);
    Handle s(L.Dget_space(d.id), L.Sclose);
    if (t.id < 0 || s.id < 0)
        throw std::runtime_error(what + ": cannot query dataset " + name);
    const size_t rec = L.Tget_size(t.id);
    const hssize_t n = L.Sget_simple_extent_npoints(s.id);
    if (rec == 0 || n < 0)
        throw std::runtime_error(what + ": bad extent of dataset " + name);
    count = (size_t)n;
    bytes.assign(rec * count + 1, 0);
    if (count > 0 && L.Dread(d.id, t.id, 0 /* H5S_ALL */, 0, 0 /* H5P_DEFAULT */, bytes.data()) < 0)
        throw std::runtime_error(what + ": cannot read dataset " + name);
    return rec;
}

int64_t le64(const uint8_t *p) {
    int64_t v;
    memcpy(&v, p, 8);
    return v;
}

} // namespace

Image readHdf5Hal(const std::string &path) {
    H5 &L = h5();
    Handle file(L.Fopen(path.c_str(), 0 /* H5F_ACC_RDONLY */, 0), L.Fclose);
    if (file.id < 0)
        throw std::runtime_error("Unable to open " + path);
    Image img;
    {
        // hdf5Alignment.cpp:189-191, 564-573 and halCommon.h:31-34: the integer part of the format version must match the
        // API's (2); a file without the attribute reads as version "0.0"
        std::string version = "0.0";
        if (L.Lexists(file.id, "Verison", 0) > 0) {
            Handle vg(L.Gopen2(file.id, "Verison", 0), L.Gclose);
            if (vg.id >= 0 && L.Aexists(vg.id, "Verison") > 0)
                version = readStringAttr(L, vg.id, "Verison", path);
        }
        if ((int)atof(version.c_str()) != 2)
            throw std::runtime_error("HAL API v2.2 incompatible with format v" + version + " HAL file.");
    }
    {
        Handle tree(L.Gopen2(file.id, "Phylogeny", 0), L.Gclose);
        if (tree.id < 0)
            throw std::runtime_error(path + ": not a HAL file (no Phylogeny group)");
        img.newick = readStringAttr(L, tree.id, "Phylogeny", path);
    }
    NewickParser np(img.newick);
    const int rootNode = np.parse();
    // genome ids: breadth first from the root, children in Newick order
    std::vector<int> order;
    std::map<int, int> idOfNode;
    {
        std::deque<int> q{rootNode};
        while (!q.empty()) {
            const int nd = q.front();
            q.pop_front();
            idOfNode[nd] = (int)order.size();
            order.push_back(nd);
            for (int k : np.nodes[(size_t)nd].kids)
                q.push_back(k);
        }
    }
    img.genomes.resize(order.size());
    for (size_t g = 0; g < order.size(); ++g) {
        const NewickNode &nd = np.nodes[(size_t)order[g]];
        GenomeTables &G = img.genomes[g];
        G.name = nd.label;
        for (int k : nd.kids) {
            G.children.push_back(idOfNode[k]);
            img.genomes[(size_t)idOfNode[k]].parent = (int)g;
            img.genomes[(size_t)idOfNode[k]].branchLength = np.nodes[(size_t)k].len;
        }
    }
    // genome groups hang off the file root (Hdf5Genome is constructed with h5Parent = the file, hdf5Alignment.cpp:422;
    // the "Genomes" group the file also contains stays empty)
    const hid_t genomesId = file.id;
    for (GenomeTables &G : img.genomes) {
        const std::string what = path + ": genome " + G.name;
        if (L.Lexists(genomesId, G.name.c_str(), 0) <= 0)
            throw std::runtime_error(path + ": genome '" + G.name + "' of the tree has no genome group");
        Handle grp(L.Gopen2(genomesId, G.name.c_str(), 0), L.Gclose);
        if (grp.id < 0)
            throw std::runtime_error(what + ": cannot open group");
        std::vector<uint8_t> raw;
        size_t n = 0;
        // sequences
        if (L.Lexists(grp.id, "SEQIDX_ARRAY", 0) > 0) {
            const size_t rec = readDataset(L, grp.id, "SEQIDX_ARRAY", raw, n, what);
            if (rec != 24 || n == 0)
                throw std::runtime_error(what + ": unexpected SEQIDX_ARRAY record size " + std::to_string(rec));
            std::vector<uint8_t> names;
            size_t nn = 0;
            const size_t nrec = readDataset(L, grp.id, "SEQNAME_ARRAY", names, nn, what);
            if (nn + 1 != n)
                throw std::runtime_error(what + ": SEQNAME_ARRAY and SEQIDX_ARRAY disagree");
            G.seqs.resize(nn);
            for (size_t s = 0; s < nn; ++s) {
                const uint8_t *r = raw.data() + 24 * s, *nx = r + 24;
                SeqInfo &S = G.seqs[s];
                S.start = le64(r);
                S.length = le64(nx) - S.start;
                S.topStart = le64(r + 8);
                S.numTop = le64(nx + 8) - S.topStart;
                S.botStart = le64(r + 16);
                S.numBot = le64(nx + 16) - S.botStart;
                const char *nm = (const char *)names.data() + nrec * s;
                S.name.assign(nm, strnlen(nm, nrec));
            }
            G.totalLength = le64(raw.data() + 24 * nn);
        }
        // top segments
        if (L.Lexists(grp.id, "TOP_ARRAY", 0) > 0) {
            const size_t rec = readDataset(L, grp.id, "TOP_ARRAY", raw, n, what);
            if (rec != 33)
                throw std::runtime_error(what + ": unexpected TOP_ARRAY record size " + std::to_string(rec));
            G.numTop = n > 0 ? (int64_t)n - 1 : 0;
        } else {
            n = 0;
            G.numTop = 0;
        }
        const size_t nt = (size_t)G.numTop;
        G.tStart.assign(nt + 1, G.totalLength);
        G.tParent.resize(nt);
        G.tParalogy.resize(nt);
        G.tBotParse.resize(nt);
        G.tParentRev.resize(nt);
        for (size_t i = 0; i < nt + 1 && n > 0; ++i) {
            const uint8_t *r = raw.data() + 33 * i;
            G.tStart[i] = le64(r);
            if (i < nt) {
                G.tBotParse[i] = le64(r + 8);
                G.tParalogy[i] = le64(r + 16);
                G.tParent[i] = le64(r + 24);
                G.tParentRev[i] = r[32] ? 1 : 0;
            }
        }
        // bottom segments
        const size_t nc = G.children.size();
        size_t brec = 0;
        if (L.Lexists(grp.id, "BOTTOM_ARRAY", 0) > 0) {
            brec = readDataset(L, grp.id, "BOTTOM_ARRAY", raw, n, what);
            if (brec < 24 || (brec - 24) % 9 != 0)
                throw std::runtime_error(what + ": unexpected BOTTOM_ARRAY record size " + std::to_string(brec));
            G.numBot = n > 0 ? (int64_t)n - 1 : 0;
            if (G.numBot > 0 && (brec - 24) / 9 != nc) // numChildrenFromDataType, hdf5BottomSegment.cpp:30-32
                throw std::runtime_error(what + ": BOTTOM_ARRAY holds " + std::to_string((brec - 24) / 9) + " child slots, the tree " +
                                         std::to_string(nc));
        } else {
            n = 0;
            G.numBot = 0;
        }
        const size_t nb = (size_t)G.numBot;
        G.bStart.assign(nb + 1, G.totalLength);
        G.bTopParse.resize(nb);
        G.bChild.assign(nc, std::vector<int64_t>(nb));
        G.bChildRev.assign(nc, std::vector<uint8_t>(nb));
        for (size_t i = 0; i < nb + 1 && n > 0; ++i) {
            const uint8_t *r = raw.data() + brec * i;
            G.bStart[i] = le64(r);
            if (i < nb) {
                G.bTopParse[i] = le64(r + 16);
                for (size_t k = 0; k < nc; ++k) {
                    G.bChild[k][i] = le64(r + 24 + 9 * k);
                    G.bChildRev[k][i] = r[24 + 9 * k + 8] ? 1 : 0;
                }
            }
        }
        // DNA: ceil(len / 2) packed bytes (hdf5Genome.cpp:115-123 rounds the dataset up for odd lengths)
        const size_t dnaBytes = ((size_t)G.totalLength + 1) / 2;
        G.dna.assign(dnaBytes, 0x44); // 'n','n' when the file carries no DNA
        if (L.Lexists(grp.id, "DNA_ARRAY", 0) > 0) {
            const size_t rec = readDataset(L, grp.id, "DNA_ARRAY", raw, n, what);
            if (rec != 1)
                throw std::runtime_error(what + ": unexpected DNA_ARRAY element size");
            if (n < dnaBytes)
                throw std::runtime_error(what + ": DNA_ARRAY shorter than the genome");
            memcpy(G.dna.data(), raw.data(), dnaBytes);
        }
    }
    return img;
}

} // namespace hgx
