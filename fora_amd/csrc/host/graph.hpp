// graph.hpp -- text graph loader of the `fora` command line.
// Same inputs and semantics as the reference's Graph (graph.h:37-163): attribute.txt gives n
// and m ("n=...", "m=..."), graph.txt is a "src dst" edge list; ids must be < n, self loops
// are dropped, duplicates kept, per-node neighbour order is file order.  Storage is CSR
// (row_ptr int64[n+1], col int32[nnz]) instead of vector<vector<int>>.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace forahost {

struct Graph {
    int32_t n = 0;
    long long m = 0; // from attribute.txt, never recounted (graph.h:58-63)
    std::vector<int64_t> row_ptr;
    std::vector<int32_t> col;
    std::string data_folder;
    std::string error;

    bool init_nm() { // graph.h:48-64
        const std::string f = data_folder + "attribute.txt";
        FILE *fp = fopen(f.c_str(), "r");
        if (!fp) { error = "attribute file " + f + " not find"; return false; } // config.cpp:20-26
        int c;
        long long v;
        while ((c = fgetc(fp)) != EOF && c != '=') {}
        if (c == EOF || fscanf(fp, "%lld", &v) != 1) { fclose(fp); error = "bad attribute file"; return false; }
        n = (int32_t)v;
        while ((c = fgetc(fp)) != EOF && c != '=') {}
        if (c == EOF || fscanf(fp, "%lld", &v) != 1) { fclose(fp); error = "bad attribute file"; return false; }
        m = v;
        fclose(fp);
        return n > 0;
    }

    bool init_graph() { // graph.h:89-163, plain branch :151-161
        if (!init_nm()) return false;
        const std::string f = data_folder + "graph.txt";
        FILE *fp = fopen(f.c_str(), "rb");
        if (!fp) { error = "graph file " + f + " not find"; return false; }
        std::vector<int32_t> src, dst;
        src.reserve((size_t)(m > 0 ? m : 0));
        dst.reserve((size_t)(m > 0 ? m : 0));
        // same token stream as `fscanf("%d%d")` (graph.h:152-154) -- whitespace-separated decimal integers, stop
        // at the first thing that is not one -- read in 16 MiB blocks instead of one libc call per number
        std::vector<char> buf(16u << 20);
        long long cur = 0, first = 0;
        bool in_num = false, neg = false, have_first = false, bad = false, stop = false;
        auto flush_num = [&]() {
            const long long v = neg ? -cur : cur;
            if (!have_first) { first = v; have_first = true; return; }
            have_first = false;
            if (first >= n || v >= n || first < 0 || v < 0) { bad = true; return; } // assert(t1 < n); assert(t2 < n);
            if (first == v) return;                                                 // graph.h:157
            src.push_back((int32_t)first);
            dst.push_back((int32_t)v);
        };
        size_t got;
        while (!stop && !bad && (got = fread(buf.data(), 1, buf.size(), fp)) > 0) {
            for (size_t i = 0; i < got && !bad; i++) {
                const char ch = buf[i];
                if (ch >= '0' && ch <= '9') { cur = in_num ? cur * 10 + (ch - '0') : (ch - '0'); in_num = true; }
                else if (ch == '-' && !in_num) { neg = true; in_num = true; cur = 0; }
                else if (ch == ' ' || ch == '\n' || ch == '\t' || ch == '\r') { if (in_num) { flush_num(); in_num = false; neg = false; } }
                else { stop = true; break; } // fscanf would fail to match here
            }
        }
        if (in_num && !bad) flush_num();
        fclose(fp);
        if (bad) { error = "node id out of range in graph.txt"; return false; }
        row_ptr.assign((size_t)n + 1, 0);
        for (int32_t s : src) row_ptr[(size_t)s + 1]++;
        for (int32_t v = 0; v < n; v++) row_ptr[(size_t)v + 1] += row_ptr[v];
        col.resize(src.size());
        std::vector<int64_t> fill(row_ptr.begin(), row_ptr.end() - 1);
        for (size_t e = 0; e < src.size(); e++) col[(size_t)fill[src[e]]++] = dst[e];
        return true;
    }

    bool load_ss_query(std::vector<int32_t> &queries) { // algo.h:511-522
        const std::string f = data_folder + "ssquery.txt";
        FILE *fp = fopen(f.c_str(), "r");
        if (!fp) { error = "query file does not exist, please generate ss query files first"; return false; }
        int v;
        while (fscanf(fp, "%d", &v) == 1) queries.push_back(v);
        fclose(fp);
        return true;
    }
};

// ---- walk index files ---------------------------------------------------------------
// Names follow build.h:147-181.  Default container: a native one with the reference's two arrays
// (rw_idx, rw_idx_info).  The reference's own files are Boost binary_oarchive streams: they are
// READ when found (tail-anchored parse, see below) and can be written with `build --boost_idx`;
// both in a layout stated from the Boost sources' documented behaviour, unpinned (no Boost here).
struct IndexFile {
    static constexpr uint64_t MAGIC = 0x31584449414F46ull; // "FOAIDX1"
    static std::string base(const std::string &loc, double rmax_scale, bool opt, const char *kind) {
        std::string f = loc + "randwalks.";
        if (rmax_scale != 1) f += std::to_string(rmax_scale) + ".";
        f += kind;
        if (opt) f += ".onehopopt";
        return f;
    }
    static bool write(const std::string &loc, double rmax_scale, bool opt, int32_t n, const std::vector<int32_t> &rw,
                      const std::vector<uint64_t> &off, const std::vector<uint64_t> &cnt, std::string &err) {
        FILE *a = fopen(base(loc, rmax_scale, opt, "idx").c_str(), "wb");
        FILE *b = fopen(base(loc, rmax_scale, opt, "info").c_str(), "wb");
        if (!a || !b) { err = "cannot write index files in " + loc; if (a) fclose(a); if (b) fclose(b); return false; }
        uint64_t h[2] = {MAGIC, (uint64_t)rw.size()};
        fwrite(h, 8, 2, a);
        fwrite(rw.data(), 4, rw.size(), a);
        uint64_t g[2] = {MAGIC, (uint64_t)n};
        fwrite(g, 8, 2, b);
        for (int32_t v = 0; v < n; v++) { uint64_t p[2] = {off[v], cnt[v]}; fwrite(p, 8, 2, b); } // pair<u64 off, ulong cnt>
        fclose(a); fclose(b);
        return true;
    }
    // ---- Boost binary_oarchive files of the reference (serialize_idx / deserialize_idx, build.h:194-217) ----------
    // Layout as far as it can be stated without Boost at hand (UNPINNED: no reference-written file was available):
    //   u64 22, "serialization::archive", u16 library version, u8 sizeof(int,long,float,double), u32 1   (40 bytes)
    //   randwalks.idx : vector<int>                -> u64 count, count raw int32                (array optimisation)
    //   randwalks.info: vector<pair<ull, ulong>>   -> class preamble (class id i16, tracking u8, version u32),
    //                                                 u64 count, count raw 16-byte pairs        (bitwise serializable)
    // The READER does not depend on the preamble: it checks the signature, takes the payload from the END of the file
    // (16 n bytes / 4 * sum(cnt) bytes) and requires the u64 in front of it to be the element count.
    static constexpr const char *BOOST_SIG = "serialization::archive";
    static bool boost_header_ok(const std::vector<unsigned char> &f) {
        if (f.size() < 40) return false;
        uint64_t len;
        memcpy(&len, f.data(), 8);
        return len == 22 && memcmp(f.data() + 8, BOOST_SIG, 22) == 0;
    }
    static bool slurp(const std::string &path, std::vector<unsigned char> &out) {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) return false;
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        out.resize((size_t)sz);
        const bool ok = sz == 0 || fread(out.data(), 1, (size_t)sz, f) == (size_t)sz;
        fclose(f);
        return ok;
    }
    static bool read_boost(const std::string &fa, const std::string &fb, int32_t n, std::vector<int32_t> &rw,
                           std::vector<uint64_t> &off, std::vector<uint64_t> &cnt, std::string &err) {
        std::vector<unsigned char> info, idx;
        if (!slurp(fb, info) || !boost_header_ok(info)) return false;
        const size_t pay = (size_t)n * 16;
        uint64_t c = 0;
        if (info.size() < 40 + 8 + pay) { err = fb + ": Boost archive too short for n pairs"; return false; }
        memcpy(&c, info.data() + info.size() - pay - 8, 8);
        if (c != (uint64_t)n) { err = fb + ": Boost archive whose element count is not n"; return false; }
        off.resize((size_t)n); cnt.resize((size_t)n);
        uint64_t total = 0;
        for (int32_t v = 0; v < n; v++) {
            uint64_t p[2];
            memcpy(p, info.data() + info.size() - pay + (size_t)v * 16, 16);
            off[v] = p[0]; cnt[v] = p[1];
            if (p[0] != total) { err = fb + ": offsets are not the running sum of the counts (build.h:331-333)"; return false; }
            total += p[1];
        }
        if (!slurp(fa, idx) || !boost_header_ok(idx)) { err = fa + ": not a Boost binary archive"; return false; }
        if (idx.size() < 40 + 8 + total * 4) { err = fa + ": Boost archive too short for the walks of its .info"; return false; }
        memcpy(&c, idx.data() + idx.size() - total * 4 - 8, 8);
        if (c != total) { err = fa + ": element count does not match its .info"; return false; }
        rw.resize(total);
        if (total) memcpy(rw.data(), idx.data() + idx.size() - total * 4, total * 4);
        for (uint64_t i = 0; i < total; i++)
            if (rw[i] < 0 || rw[i] >= n) { err = fa + ": walk endpoint out of range"; return false; }
        return true;
    }
    // Writer in the same presumed layout (`fora build --boost_idx`); library version 15 = Boost 1.64-1.65 (README.md:46)
    static bool write_boost(const std::string &loc, double rmax_scale, bool opt, int32_t n, const std::vector<int32_t> &rw,
                            const std::vector<uint64_t> &off, const std::vector<uint64_t> &cnt, std::string &err) {
        FILE *a = fopen(base(loc, rmax_scale, opt, "idx").c_str(), "wb");
        FILE *b = fopen(base(loc, rmax_scale, opt, "info").c_str(), "wb");
        if (!a || !b) { err = "cannot write index files in " + loc; if (a) fclose(a); if (b) fclose(b); return false; }
        auto header = [](FILE *f) {
            const uint64_t len = 22;
            const uint16_t ver = 15;
            const unsigned char sizes[4] = {4, 8, 4, 8};
            const uint32_t one = 1;
            fwrite(&len, 8, 1, f); fwrite(BOOST_SIG, 1, 22, f); fwrite(&ver, 2, 1, f); fwrite(sizes, 1, 4, f); fwrite(&one, 4, 1, f);
        };
        header(a);
        const uint64_t na = rw.size();
        fwrite(&na, 8, 1, a);
        fwrite(rw.data(), 4, rw.size(), a);
        header(b);
        const int16_t class_id = 0; const unsigned char tracking = 0; const uint32_t version = 0;
        fwrite(&class_id, 2, 1, b); fwrite(&tracking, 1, 1, b); fwrite(&version, 4, 1, b);
        const uint64_t nb = (uint64_t)n;
        fwrite(&nb, 8, 1, b);
        for (int32_t v = 0; v < n; v++) { uint64_t p[2] = {off[v], cnt[v]}; fwrite(p, 8, 2, b); }
        fclose(a); fclose(b);
        return true;
    }
    static bool read(const std::string &loc, double rmax_scale, bool opt, int32_t n, std::vector<int32_t> &rw,
                     std::vector<uint64_t> &off, std::vector<uint64_t> &cnt, std::string &err) {
        const std::string fa = base(loc, rmax_scale, opt, "idx"), fb = base(loc, rmax_scale, opt, "info");
        FILE *a = fopen(fa.c_str(), "rb");
        if (!a) { err = "index file " + fa + " not find"; return false; } // assert_file_exist, build.h:197
        FILE *b = fopen(fb.c_str(), "rb");
        if (!b) { fclose(a); err = "index file " + fb + " not find"; return false; }
        uint64_t h[2], g[2];
        bool ok = fread(h, 8, 2, a) == 2 && fread(g, 8, 2, b) == 2 && h[0] == MAGIC && g[0] == MAGIC && g[1] == (uint64_t)n;
        if (ok) {
            rw.resize(h[1]);
            ok = fread(rw.data(), 4, rw.size(), a) == rw.size();
            off.resize((size_t)n); cnt.resize((size_t)n);
            for (int32_t v = 0; ok && v < n; v++) { uint64_t p[2]; ok = fread(p, 8, 2, b) == 2; off[v] = p[0]; cnt[v] = p[1]; }
        }
        fclose(a); fclose(b);
        if (!ok) { // not this build's container: a Boost archive written by the reference?
            std::string berr;
            if (read_boost(fa, fb, n, rw, off, cnt, berr)) return true;
            err = berr.empty() ? "index files are neither this build's container nor a Boost binary archive" : berr;
        }
        return ok;
    }
};

} // namespace forahost
