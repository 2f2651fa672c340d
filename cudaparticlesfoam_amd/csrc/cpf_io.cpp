// Output path: D2H + ASCII .vtu writer with the reference's on-disk layout.
//
// Restates the format of writeParticles2VTU (third_party/RTXAdvect/cuda/utils.cpp:144-283):
// one VTK_VERTEX cell per particle, arrays Position (Float64, "%.15lf"), ParticleType (w),
// ParticleID, ParticleTetID, ConvexTetID, vels ("%lf", NaN -> 0), KEs, then connectivity /
// offsets / types.  Differences, on purpose:
//   * ParticleTetID / ConvexTetID both hold the containing CELL id (there are no tets here; the
//     reference's ParticleTetID is uninitialised memory in its default ConvexPoly build);
//   * KEs holds the kinetic energy (the reference prints 0 for every non-zero KE);
//   * no system("pause") on NaN: the function returns CPF_WARN_NAN instead (the file is written).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "cpf.h"

struct cpf_context;
namespace cpf { bool vtu_binary(const cpf_context* ctx); }     // cpf_api.cpp: the context's "vtu_binary" option

namespace {

// Formatting dominates the output path (printf of "%.15lf": 2.4 us per particle and frame on one core, 240 ms for
// the tutorial's 1e5 particles against 10 ms of GPU time for the 1000 cycles between two frames), so every
// DataArray is formatted in parallel: chunks of 8192 particles, one std::thread each (at most 64, or the machine's cores), the
// texts written in order.  The bytes are those of the serial loop.  (Round 5, a 256-core box: the tutorial's 1e5 particles are
// out after 9-16 ms -- 22-23 ms with round 3's 32768 x 16 in barrier-separated groups --, TJunction's 4e6 after 140-240 ms, was
// 250-580.)
#ifndef CPF_WRITER_CHUNK
#define CPF_WRITER_CHUNK 8192
#endif
constexpr long long kChunk = CPF_WRITER_CHUNK;

int writer_threads(long long n) {
    const unsigned hw = std::thread::hardware_concurrency();
    const long long want = (n + kChunk - 1) / kChunk;
#ifndef CPF_WRITER_MAX_THREADS
#define CPF_WRITER_MAX_THREADS 64
#endif
    return (int)std::max<long long>(1, std::min<long long>({want, (long long)(hw ? hw : 1), CPF_WRITER_MAX_THREADS}));
}

// ---- "%.Nlf" without printf.  glibc prints the EXACT decimal expansion of the double, rounded half-to-even at the
// last digit; so does this: x = m * 2^e with a 53-bit m, so m * 10^N (N <= 15) fits 128 bits, and the shift by e is
// either exact (e >= 0) or a division by a power of two whose remainder decides the rounding.  Anything that is
// not finite, or too large for the 128-bit product, goes through snprintf.  Checked against correctly rounded formatting on
// 1.2 million doubles of every magnitude and the edge cases (tests/test_vtu_writer.py).  ~10x faster than printf.
typedef unsigned __int128 u128;

// "00" "01" ... "99"
struct DigitPairs {
    char t[200];
    constexpr DigitPairs() : t() { for (int i = 0; i < 100; ++i) { t[2 * i] = (char)('0' + i / 10); t[2 * i + 1] = (char)('0' + i % 10); } }
};
constexpr DigitPairs kDigitPairTable{};
constexpr const char* kDigitPairs = kDigitPairTable.t;

inline char* put_u64(char* p, unsigned long long v) {
    if (v < 10) { *p++ = (char)('0' + v); return p; }          // (coordinates and velocities of order one: the usual case)
    char tmp[24]; int k = 0;
    while (v >= 100) { const unsigned d = (unsigned)(v % 100); v /= 100; tmp[k++] = kDigitPairs[2 * d + 1]; tmp[k++] = kDigitPairs[2 * d]; }
    if (v >= 10) { tmp[k++] = kDigitPairs[2 * v + 1]; tmp[k++] = kDigitPairs[2 * v]; }
    else tmp[k++] = (char)('0' + v);
    while (k) *p++ = tmp[--k];
    return p;
}

template <int N>   // N fractional digits (6 or 15)
inline void append_fixed(std::string& out, double x) {
    static_assert(N == 6 || N == 15, "");
    constexpr unsigned long long P10 = N == 6 ? 1000000ull : 1000000000000000ull;
    unsigned long long bits; std::memcpy(&bits, &x, 8);
    const int be = (int)((bits >> 52) & 0x7ff);
    if (be == 0x7ff || be >= 1023 + 63) {                 // inf / nan / |x| >= 2^63: the library's own text
        char b[400];
        const int k = std::snprintf(b, sizeof(b), N == 6 ? "%lf" : "%.15lf", x);
        out.append(b, (size_t)k);
        return;
    }
    const bool neg = (bits >> 63) != 0;
    unsigned long long m = bits & ((1ull << 52) - 1);
    int e;
    if (be == 0) e = -1074; else { m |= 1ull << 52; e = be - 1075; }      // x = m * 2^e
    u128 v = (u128)m * P10;                                                // < 2^53 * 2^50
    u128 q;
    if (e >= 0) q = v << e;                                                // |x| < 2^63 and 10^15 < 2^50: no overflow
    else if (-e >= 128) q = 0;                                             // v < 2^103 <= half of 2^-e: rounds to 0
    else {
        const int sh = -e;
        q = v >> sh;
        const u128 rem = v & (((u128)1 << sh) - 1), half = (u128)1 << (sh - 1);
        if (rem > half || (rem == half && (q & 1))) ++q;                    // round half to even, like glibc
    }
    unsigned long long ip, fp;
    if ((unsigned long long)(q >> 64) == 0) {                               // |x| < 18446 (15 digits): plain 64-bit division
        const unsigned long long q64 = (unsigned long long)q;
        ip = q64 / P10; fp = q64 % P10;
    } else {
        ip = (unsigned long long)(q / P10); fp = (unsigned long long)(q % P10);
    }
    char b[48]; char* p = b;
    if (neg) *p++ = '-';
    p = put_u64(p, ip);
    *p++ = '.';
    // the N fractional digits, two at a time out of a table (half the divisions of the digit-by-digit loop)
    int k = N;
    for (; k >= 2; k -= 2) { const unsigned d = (unsigned)(fp % 100); fp /= 100; p[k - 2] = kDigitPairs[2 * d]; p[k - 1] = kDigitPairs[2 * d + 1]; }
    if (k == 1) p[0] = (char)('0' + (fp % 10));
    p += N;
    out.append(b, (size_t)(p - b));
}

inline void append_int(std::string& out, long long v) {
    char b[24]; char* p = b;
    unsigned long long u = v < 0 ? (unsigned long long)(-(v + 1)) + 1ull : (unsigned long long)v;
    if (v < 0) *p++ = '-';
    p = put_u64(p, u);
    *p++ = '\n';
    out.append(b, (size_t)(p - b));
}

template <int N>
inline void append_fixed3(std::string& out, double a, double b, double c) {
    append_fixed<N>(out, a); out.push_back(' ');
    append_fixed<N>(out, b); out.push_back(' ');
    append_fixed<N>(out, c); out.push_back('\n');
}

// item(i, buf) appends the text of element i to buf.  All chunks of the array are formatted first -- T threads taking chunk
// numbers from a counter, each chunk into its own string (no barrier between groups of chunks, T thread starts per array
// instead of T per group) -- and then written in order.  `bytesPerItem`: the size the strings are reserved with.
template <typename F>
bool write_array(FILE* fp, long long n, int bytesPerItem, const F& item) {
    const int T = writer_threads(n);
    const long long nChunks = (n + kChunk - 1) / kChunk;
    std::vector<std::string> text((size_t)nChunks);
    std::atomic<long long> next{0};
    auto work = [&] {
        for (;;) {
            const long long c = next.fetch_add(1, std::memory_order_relaxed);
            if (c >= nChunks) return;
            std::string& out = text[(size_t)c];
            const long long i0 = c * kChunk, i1 = std::min(n, i0 + kChunk);
            out.reserve((size_t)(i1 - i0) * (size_t)bytesPerItem);
            for (long long i = i0; i < i1; ++i) item(i, out);
        }
    };
    if (T == 1) work();
    else {
        std::vector<std::thread> pool;
        for (int t = 1; t < T; ++t) pool.emplace_back(work);
        work();
        for (auto& th : pool) th.join();
    }
    for (const std::string& s : text)
        if (!s.empty() && std::fwrite(s.data(), 1, s.size(), fp) != s.size()) return false;
    return true;
}

}  // namespace

extern "C" int cpf_write_vtu_arrays(const char* path, int64_t n, const double* xyzw, const int32_t* cell,
                                    const double* vel, double* totalKE) {
    if (!path || n < 0 || (n > 0 && (!xyzw || !cell || !vel))) return CPF_ERR_ARG;
    FILE* fp = std::fopen(path, "w");
    if (!fp) return CPF_ERR_ARG;
    const long long N = (long long)n;
    bool ok = true;
    std::fprintf(fp, "<VTKFile type='UnstructuredGrid' version='1.0' byte_order='LittleEndian' header_type='UInt64'>\n");
    std::fprintf(fp, "<UnstructuredGrid>\n");
    std::fprintf(fp, "<Piece NumberOfCells='%lld' NumberOfPoints='%lld'>\n", N, N);
    std::fprintf(fp, "<Points>\n");
    std::fprintf(fp, "<DataArray NumberOfComponents='3' type='Float64' Name='Position' format='ascii'>\n");
    ok &= write_array(fp, N, 64, [&](long long i, std::string& o) {
        append_fixed3<15>(o, xyzw[4 * i], xyzw[4 * i + 1], xyzw[4 * i + 2]); });
    std::fprintf(fp, "</DataArray>\n</Points>\n<PointData>\n");
    std::fprintf(fp, "<DataArray NumberOfComponents='1' type='Int32' Name='ParticleType' format='ascii'>\n");
    ok &= write_array(fp, N, 4, [&](long long i, std::string& o) { append_int(o, (int)xyzw[4 * i + 3]); });
    std::fprintf(fp, "</DataArray>\n");
    std::fprintf(fp, "<DataArray NumberOfComponents='1' type='Int32' Name='ParticleID' format='ascii'>\n");
    ok &= write_array(fp, N, 12, [&](long long i, std::string& o) { append_int(o, i); });
    std::fprintf(fp, "</DataArray>\n");
    for (const char* name : {"ParticleTetID", "ConvexTetID"}) {
        std::fprintf(fp, "<DataArray NumberOfComponents='1' type='Int32' Name='%s' format='ascii'>\n", name);
        ok &= write_array(fp, N, 12, [&](long long i, std::string& o) { append_int(o, cell[i]); });
        std::fprintf(fp, "</DataArray>\n");
    }
    std::fprintf(fp, "<DataArray NumberOfComponents='3' type='Float32' Name='vels' format='ascii'>\n");
    ok &= write_array(fp, N, 40, [&](long long i, std::string& o) {
        if (std::isnan(vel[4 * i])) append_fixed3<6>(o, 0.0, 0.0, 0.0);
        else append_fixed3<6>(o, vel[4 * i], vel[4 * i + 1], vel[4 * i + 2]); });
    std::fprintf(fp, "</DataArray>\n");
    std::fprintf(fp, "<DataArray NumberOfComponents='1' type='Float32' Name='KEs' format='ascii'>\n");
    auto keOf = [&](long long i) {
        return 0.5 * (vel[4 * i] * vel[4 * i] + vel[4 * i + 1] * vel[4 * i + 1] + vel[4 * i + 2] * vel[4 * i + 2]); };
    ok &= write_array(fp, N, 16, [&](long long i, std::string& o) { append_fixed<6>(o, keOf(i)); o.push_back('\n'); });
    double total = 0.0;
    for (long long i = 0; i < N; ++i) total += keOf(i);              // in index order, like the reference's running sum
    std::fprintf(fp, "</DataArray>\n</PointData>\n<Cells>\n");
    std::fprintf(fp, "<DataArray type='Int32' Name='connectivity' format='ascii'>\n");
    ok &= write_array(fp, N, 12, [&](long long i, std::string& o) { append_int(o, i); });
    std::fprintf(fp, "</DataArray>\n<DataArray type='Int32' Name='offsets' format='ascii'>\n");
    ok &= write_array(fp, N, 12, [&](long long i, std::string& o) { append_int(o, i + 1); });
    std::fprintf(fp, "</DataArray>\n<DataArray type='UInt8' Name='types' format='ascii'>\n");
    ok &= write_array(fp, N, 2, [&](long long, std::string& o) { o.append("1\n"); });
    std::fprintf(fp, "</DataArray>\n</Cells>\n</Piece>\n</UnstructuredGrid>\n</VTKFile>\n");
    const bool bad = !ok || std::ferror(fp) != 0;
    std::fclose(fp);
    if (totalKE) *totalKE = total;
    if (bad) return CPF_ERR_ARG;
    return std::isnan(total) ? CPF_WARN_NAN : CPF_OK;
}

// ---- the same frame with the arrays RAW behind the XML ("binary-appended as an option").  Same arrays, names,
// types and order as the ASCII file -- Position Float64 x 3, ParticleType / ParticleID / ParticleTetID / ConvexTetID Int32, vels
// Float32 x 3 (NaN -> 0), KEs Float32, connectivity / offsets Int32, types UInt8 -- as DataArrays with format='appended' and an
// offset into one <AppendedData encoding='raw'> section: per array a UInt64 byte count, then the bytes (little endian).  Nothing to
// format: 1e5 particles are 6 MB instead of 9 MB of text and take a memcpy's time instead of 60-80 ms.  ParaView / VTK read both.
extern "C" int cpf_write_vtu_arrays_binary(const char* path, int64_t n, const double* xyzw, const int32_t* cell,
                                           const double* vel, double* totalKE) {
    if (!path || n < 0 || n > 0x7fffffffLL || (n > 0 && (!xyzw || !cell || !vel))) return CPF_ERR_ARG;
    FILE* fp = std::fopen(path, "wb");
    if (!fp) return CPF_ERR_ARG;
    const size_t N = (size_t)n;
    struct Arr { const char* name; const char* type; int comps; size_t bytes; };
    const Arr arrs[] = {{"Position", "Float64", 3, N * 24}, {"ParticleType", "Int32", 1, N * 4}, {"ParticleID", "Int32", 1, N * 4},
                        {"ParticleTetID", "Int32", 1, N * 4}, {"ConvexTetID", "Int32", 1, N * 4}, {"vels", "Float32", 3, N * 12},
                        {"KEs", "Float32", 1, N * 4}, {"connectivity", "Int32", 1, N * 4}, {"offsets", "Int32", 1, N * 4},
                        {"types", "UInt8", 1, N}};
    unsigned long long off[10], o = 0;
    for (int k = 0; k < 10; ++k) { off[k] = o; o += 8 + arrs[k].bytes; }
    auto decl = [&](int k) {
        std::fprintf(fp, "<DataArray NumberOfComponents='%d' type='%s' Name='%s' format='appended' offset='%llu'/>\n", arrs[k].comps,
                     arrs[k].type, arrs[k].name, off[k]);
    };
    std::fprintf(fp, "<VTKFile type='UnstructuredGrid' version='1.0' byte_order='LittleEndian' header_type='UInt64'>\n");
    std::fprintf(fp, "<UnstructuredGrid>\n<Piece NumberOfCells='%lld' NumberOfPoints='%lld'>\n<Points>\n", (long long)n, (long long)n);
    decl(0);
    std::fprintf(fp, "</Points>\n<PointData>\n");
    for (int k = 1; k <= 6; ++k) decl(k);
    std::fprintf(fp, "</PointData>\n<Cells>\n");
    for (int k = 7; k <= 9; ++k) decl(k);
    std::fprintf(fp, "</Cells>\n</Piece>\n</UnstructuredGrid>\n<AppendedData encoding='raw'>\n_");
    bool ok = true;
    auto block = [&](const void* data, size_t bytes) {
        const unsigned long long nb = bytes;
        ok = ok && std::fwrite(&nb, 8, 1, fp) == 1 && (bytes == 0 || std::fwrite(data, 1, bytes, fp) == bytes);
    };
    std::vector<double> pos(N * 3);
    std::vector<int32_t> i32(N);
    std::vector<float> f32(N * 3);
    for (size_t i = 0; i < N; ++i) { pos[3 * i] = xyzw[4 * i]; pos[3 * i + 1] = xyzw[4 * i + 1]; pos[3 * i + 2] = xyzw[4 * i + 2]; }
    block(pos.data(), N * 24);
    for (size_t i = 0; i < N; ++i) i32[i] = (int32_t)xyzw[4 * i + 3];
    block(i32.data(), N * 4);
    for (size_t i = 0; i < N; ++i) i32[i] = (int32_t)i;
    block(i32.data(), N * 4);
    block(cell, N * 4); block(cell, N * 4);
    double total = 0.0;
    std::vector<float> ke(N);
    for (size_t i = 0; i < N; ++i) {
        const bool nan = std::isnan(vel[4 * i]);
        for (int c = 0; c < 3; ++c) f32[3 * i + c] = nan ? 0.0f : (float)vel[4 * i + c];
        const double e = 0.5 * (vel[4 * i] * vel[4 * i] + vel[4 * i + 1] * vel[4 * i + 1] + vel[4 * i + 2] * vel[4 * i + 2]);
        ke[i] = (float)e;
        total += e;                                                      // in index order, like the reference's running sum
    }
    block(f32.data(), N * 12);
    block(ke.data(), N * 4);
    for (size_t i = 0; i < N; ++i) i32[i] = (int32_t)i;
    block(i32.data(), N * 4);
    for (size_t i = 0; i < N; ++i) i32[i] = (int32_t)(i + 1);
    block(i32.data(), N * 4);
    std::vector<unsigned char> types(N, 1);
    block(types.data(), N);
    std::fprintf(fp, "\n</AppendedData>\n</VTKFile>\n");
    const bool bad = !ok || std::ferror(fp) != 0;
    std::fclose(fp);
    if (totalKE) *totalKE = total;
    if (bad) return CPF_ERR_ARG;
    return std::isnan(total) ? CPF_WARN_NAN : CPF_OK;
}

extern "C" int cpf_write_vtu(cpf_context* ctx, const char* path, double* totalKE) {
    if (!ctx || !path) return CPF_ERR_ARG;
    int64_t n = 0;
    int r = cpf_num_particles(ctx, &n);
    if (r) return r;
    std::vector<double> xyzw((size_t)n * 4), vel((size_t)n * 4);
    std::vector<int32_t> cell((size_t)n);
    r = cpf_get_particles(ctx, xyzw.data(), cell.data(), vel.data());
    if (r) return r;
    const bool binary = cpf::vtu_binary(ctx);              // cpf_set_option "vtu_binary"
    return (binary ? cpf_write_vtu_arrays_binary : cpf_write_vtu_arrays)(path, n, xyzw.data(), cell.data(), vel.data(), totalKE);
}
