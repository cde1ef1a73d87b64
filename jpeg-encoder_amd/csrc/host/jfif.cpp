// JFIF container writer (host side): the marker segments the reference emits
// around the scan data, byte for byte (src/jpeg/file.rs:10-137).
#include <cstdio>
#include <cstring>

#include "jpeg_host.h"
#include "tables.h"

namespace {

struct Out {
    uint8_t *p;
    size_t cap, n;
    void u8(unsigned v) {
        if (p && n < cap) p[n] = (uint8_t)v;
        n++;
    }
    void u16(unsigned v) { u8(v >> 8); u8(v & 0xff); }  // big-endian, as byteorder::BigEndian
};

void dqt(Out &o, int id, const uint8_t *table) {
    o.u16(0xffdb); o.u16(67); o.u8(id);
    for (int k = 0; k < 64; k++) o.u8(table[jpgtab::ZIGZAG[k]]);  // zigzag order, src/jpeg/file.rs:33-35
}

void dht(Out &o, int cls, int id, const uint8_t *bits, const uint8_t *vals) {
    int n = 0;
    for (int i = 0; i < 16; i++) n += bits[i];
    o.u16(0xffc4); o.u16(19 + n); o.u8((cls << 4) | id);
    for (int i = 0; i < 16; i++) o.u8(bits[i]);
    for (int i = 0; i < n; i++) o.u8(vals[i]);
}

void header(Out &o, int32_t width, int32_t height, int h, int v) {
    o.u16(0xffd8);                                           // SOI
    o.u16(0xffe0); o.u16(16);                                // APP0
    o.u8('J'); o.u8('F'); o.u8('I'); o.u8('F'); o.u8(0);
    o.u8(1); o.u8(1); o.u8(1); o.u16(72); o.u16(72); o.u8(0); o.u8(0);
    dqt(o, 0, jpgtab::QT_LUMA);
    dqt(o, 1, jpgtab::QT_CHROMA);
    o.u16(0xffc0); o.u16(17); o.u8(8);                       // SOF0; dimensions truncated `as u16`
    o.u16((unsigned)(uint16_t)height); o.u16((unsigned)(uint16_t)width);
    o.u8(3);
    o.u8(1); o.u8((h << 4) | v); o.u8(0);
    o.u8(2); o.u8(0x11); o.u8(1);
    o.u8(3); o.u8(0x11); o.u8(1);
    dht(o, 0, 0, jpgtab::DC_LUMA_BITS, jpgtab::DC_VALS);
    dht(o, 0, 1, jpgtab::DC_CHROMA_BITS, jpgtab::DC_VALS);
    dht(o, 1, 0, jpgtab::AC_LUMA_BITS, jpgtab::AC_LUMA_VALS);
    dht(o, 1, 1, jpgtab::AC_CHROMA_BITS, jpgtab::AC_CHROMA_VALS);
    o.u16(0xffda); o.u16(12); o.u8(3);                       // SOS
    o.u8(1); o.u8(0x00); o.u8(2); o.u8(0x11); o.u8(3); o.u8(0x11);
    o.u8(0); o.u8(63); o.u8(0);
}

}  // namespace

extern "C" size_t jph_jfif_header(uint8_t *out, size_t cap, int32_t width, int32_t height, int h, int v) {
    Out o{out, cap, 0};
    header(o, width, height, h, v);
    return o.n;
}

extern "C" size_t jph_jpeg_assemble(uint8_t *out, size_t cap, int32_t width, int32_t height, int h, int v,
                                    const uint8_t *scan, size_t scan_len) {
    Out o{out, cap, 0};
    header(o, width, height, h, v);
    if (out && o.n + scan_len <= cap && scan_len) memcpy(out + o.n, scan, scan_len);
    o.n += scan_len;
    o.u16(0xffd9);
    return o.n;
}

extern "C" int jph_jpeg_write(const char *path, int32_t width, int32_t height, int h, int v, const uint8_t *scan,
                              size_t scan_len) {
    if (!path || (!scan && scan_len)) return JPH_ERR_ARG;
    uint8_t hdr[1024];
    const size_t n = jph_jfif_header(hdr, sizeof hdr, width, height, h, v);
    FILE *f = fopen(path, "wb");
    if (!f) return JPH_ERR_OPEN;
    static const uint8_t eoi[2] = {0xff, 0xd9};
    bool ok = fwrite(hdr, 1, n, f) == n;
    ok = ok && (scan_len == 0 || fwrite(scan, 1, scan_len, f) == scan_len);
    ok = ok && fwrite(eoi, 1, 2, f) == 2;
    ok = (fclose(f) == 0) && ok;
    return ok ? JPH_OK : JPH_ERR_IO;
}
