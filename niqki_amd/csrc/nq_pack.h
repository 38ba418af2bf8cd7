// nq_pack.h -- "packed FASTA": fewer bytes per base across PCIe, the raw file bytes back on the device.
//
// The reference reads its files line by line on the host (Index::Biogetline, src/niqki_index.cpp:890-941, under
// `omp critical(input)`, :391-394); here the host only moves file bytes and the GPU frames the records
// (nq_ingest.hip).  Whole genomes as FASTA files are then bound by the host-to-device copy: 1 byte per base.
// A FASTA file is almost entirely full lines of one width holding only A, C, G, T; such a run of lines is a
// PERIODIC segment and travels as 2 bits per base, lines byte aligned (70 bases + '\n' -> 18 bytes); everything
// else -- header lines, lines with any other byte (N, lower case, '\r'), a last line without '\n' -- travels as RAW
// segments, verbatim.  The device side (nq::unpack_kernel) writes back EXACTLY the file's bytes, which the
// framing kernels then read as if they had been copied: nothing downstream knows of the packing, and whether a
// file was packed can never change a result.
//
// Container (little endian):   PackHeader | n_seg x PackSeg | payload
//   raw segment       count = bytes,  width = 0:  payload holds the bytes
//   periodic segment  count = lines,  width = w:  every line is w bases + '\n'; payload holds (w + 3) / 4 bytes per
//                     line, base j of a line in bits 2 (j % 4) of byte j / 4, code = (ascii >> 1) & 3
//                     (A 0, C 1, T 2, G 3)
// Host code only (no HIP): compiled into libniqki_hip.so (niqki_pack_fasta) and, for the CPU test suite, as is.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace nqp {

constexpr uint32_t kMagic = 0x4B50514Eu;   // "NQPK"

struct PackHeader {
  uint32_t magic;
  uint32_t n_seg;
  uint64_t raw_len;       // bytes of the file
  uint64_t payload_off;   // from the start of the container (16-byte aligned)
  uint64_t payload_len;
};
struct PackSeg {
  uint64_t raw_off;   // first byte of the segment in the file
  uint64_t pk_off;    // ... in the payload
  uint32_t count;     // raw: bytes; periodic: lines
  uint32_t width;     // 0 = raw; else bases per line (each line is followed by '\n')
};
static_assert(sizeof(PackHeader) == 32 && sizeof(PackSeg) == 24, "container layout");

inline uint64_t seg_raw_len(const PackSeg &s) { return s.width ? (uint64_t)s.count * (s.width + 1ull) : s.count; }
inline uint64_t seg_pk_len(const PackSeg &s) { return s.width ? (uint64_t)s.count * (((uint64_t)s.width + 3u) / 4u) : s.count; }
constexpr uint32_t kMaxWidth = 0xFFFFFFu;   // the widest line the packer writes as a periodic segment (Packer::feed)

// Packs one line of n bases (no newline) into (n + 3) / 4 bytes; false (nothing useful written) when a byte is not
// one of A C G T.
inline bool pack_line_scalar(const uint8_t *s, size_t n, uint8_t *out) {
  static const uint8_t canon[4] = {'A', 'C', 'T', 'G'};
  size_t j = 0;
  for (; j + 4 <= n; j += 4) {
    const uint32_t c0 = (s[j] >> 1) & 3u, c1 = (s[j + 1] >> 1) & 3u, c2 = (s[j + 2] >> 1) & 3u, c3 = (s[j + 3] >> 1) & 3u;
    if ((canon[c0] ^ s[j]) | (canon[c1] ^ s[j + 1]) | (canon[c2] ^ s[j + 2]) | (canon[c3] ^ s[j + 3])) return false;
    out[j >> 2] = (uint8_t)(c0 | c1 << 2 | c2 << 4 | c3 << 6);
  }
  if (j < n) {
    uint32_t b = 0;
    for (size_t k = j; k < n; ++k) {
      const uint32_t c = (s[k] >> 1) & 3u;
      if (canon[c] != s[k]) return false;
      b |= c << (2 * (k - j));
    }
    out[j >> 2] = (uint8_t)b;
  }
  return true;
}

#if defined(__x86_64__)
// 32 bases per step: codes by shift + mask, validity by comparing with the code's own letter, four codes to a byte
// by two multiply-adds.  slack: bytes readable behind the line's last base (its newline and what follows): with 3 or
// more the last (n % 32) bases are taken by one more step that overlaps the one before instead of a scalar loop.
__attribute__((target("avx2"))) inline bool pack_line_avx2(const uint8_t *s, size_t n, uint8_t *out, size_t slack) {
  const __m256i three = _mm256_set1_epi8(3);
  const __m256i letters = _mm256_setr_epi8('A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 'A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0,
                                           0, 0, 0, 0);
  const __m256i w14 = _mm256_set1_epi16(0x0401);       // byte pair (c0, c1) -> c0 + 4 c1
  const __m256i w116 = _mm256_set1_epi32(0x00100001);  // word pair (a, b)  -> a + 16 b
  const __m256i pick = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1,
                                        -1, -1, -1, -1, -1);
  __m256i bad = _mm256_setzero_si256();
  auto step = [&](const uint8_t *p, uint8_t *o, __m256i keep) __attribute__((target("avx2"))) {
    const __m256i v = _mm256_loadu_si256((const __m256i *)p);
    const __m256i c = _mm256_and_si256(_mm256_and_si256(_mm256_srli_epi16(v, 1), three), keep);
    bad = _mm256_or_si256(bad, _mm256_and_si256(_mm256_xor_si256(_mm256_shuffle_epi8(letters, c), v), keep));
    const __m256i q = _mm256_shuffle_epi8(_mm256_madd_epi16(_mm256_maddubs_epi16(c, w14), w116), pick);
    const uint32_t lo = (uint32_t)_mm256_cvtsi256_si32(q), hi = (uint32_t)_mm256_extract_epi32(q, 4);
    memcpy(o, &lo, 4);
    memcpy(o + 4, &hi, 4);
  };
  const __m256i all = _mm256_set1_epi8(-1);
  size_t j = 0;
  for (; j + 32 <= n; j += 32) step(s + j, out + (j >> 2), all);
  if (j < n) {
    const size_t js = (n - 29) & ~(size_t)3;   // last step: bases [js, js + 32), js a multiple of 4, js + 32 <= n + 3
    if (n >= 32 && slack >= 3) {
      const __m256i iota = _mm256_setr_epi8(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28,
                                            29, 30, 31);
      step(s + js, out + (js >> 2), _mm256_cmpgt_epi8(_mm256_set1_epi8((char)(n - js)), iota));   // lanes behind the line: ignored, zero bits
    } else {
      if (!_mm256_testz_si256(bad, bad)) return false;
      return pack_line_scalar(s + j, n - j, out + (j >> 2));
    }
  }
  return _mm256_testz_si256(bad, bad);
}
#endif

inline bool pack_line(const uint8_t *s, size_t n, uint8_t *out, size_t slack) {
#if defined(__x86_64__)
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2 && n >= 32) return pack_line_avx2(s, n, out, slack);
#endif
  (void)slack;
  return pack_line_scalar(s, n, out);
}

// Worst-case container size for a file of n bytes that pack() accepts (it gives up on files that would need more
// than n / 256 + 8 segments, and a line never grows).
inline size_t pack_bound(size_t n) { return sizeof(PackHeader) + (n / 256 + 16) * sizeof(PackSeg) + 16 + n; }

// The container of a file, fed piece by piece (a reader thread packs a file while it reads it: the pieces stay in its
// cache).  begin(out, cap, n): n = the file's size, cap >= pack_bound(n).  feed(buf, m, last): buf holds the file's
// next m bytes, `last` = the file ends with them; whole lines are taken, the return value says how many bytes -- the
// caller brings the rest back at the front of the next piece (with `last` everything is taken).  finish(): the
// container's size, or 0 when the file is not worth packing (too fragmented: a reads file, FASTQ; or the container
// would not be smaller than the file) -- the caller then sends the raw bytes.
struct Packer {
  uint8_t *out = nullptr;
  PackSeg *seg = nullptr;
  uint8_t *pay = nullptr;
  size_t max_seg = 0, n_seg = 0, pk = 0;
  uint64_t pos = 0, total = 0;   // raw bytes taken so far / of the whole file
  bool ok = false;

  void begin(uint8_t *out_, size_t cap, uint64_t n) {
    out = out_;
    total = n;
    n_seg = pk = 0;
    pos = 0;
    ok = n >= 64 && cap >= pack_bound(n);
    max_seg = n / 256 + 8;
    seg = (PackSeg *)(out + sizeof(PackHeader));
    // (the table is written in place at its worst-case size's start; the payload starts behind max_seg entries and is
    // moved down once the real count is known)
    pay = out + sizeof(PackHeader) + (max_seg + 1) * sizeof(PackSeg);
  }

  size_t feed(const uint8_t *raw, size_t n, bool last) {
    if (!ok) return n;
    size_t i = 0;
    while (i < n) {
      // inside a run of equal lines the next newline is where the last one was: a line that does pack holds no
      // newline (it is not A C G T), so the guess needs no search to be verified
      PackSeg *run = n_seg && seg[n_seg - 1].width ? &seg[n_seg - 1] : nullptr;
      if (run && i + run->width < n && raw[i + run->width] == '\n' && run->count < 0xFFFFFFFFu &&
          pack_line(raw + i, run->width, pay + pk, n - (i + run->width))) {
        run->count += 1;
        pk += (run->width + 3u) / 4u;
        i += run->width + 1u;
        continue;
      }
      const uint8_t *nl = (const uint8_t *)memchr(raw + i, '\n', n - i);
      if (!nl && !last) break;   // the line goes on in the next piece
      const size_t len = nl ? (size_t)(nl - (raw + i)) : n - i;   // without the newline
      bool packed = false;
      if (nl && len >= 16 && len <= kMaxWidth) {
        PackSeg *cur = n_seg ? &seg[n_seg - 1] : nullptr;
        const bool extend = cur && cur->width == len && cur->count < 0xFFFFFFFFu;
        if (pack_line(raw + i, len, pay + pk, n - (i + len))) {
          if (!extend) {
            if (n_seg == max_seg) { ok = false; return n; }
            seg[n_seg++] = PackSeg{pos + i, (uint64_t)pk, 0u, (uint32_t)len};
            cur = &seg[n_seg - 1];
          }
          cur->count += 1;
          pk += (len + 3) / 4;
          packed = true;
        }
      }
      const size_t line_bytes = len + (nl ? 1 : 0);
      if (!packed) {
        PackSeg *cur = n_seg ? &seg[n_seg - 1] : nullptr;
        if (!(cur && cur->width == 0 && (uint64_t)cur->count + line_bytes <= 0xFFFFFFFFull)) {
          if (n_seg == max_seg) { ok = false; return n; }
          seg[n_seg++] = PackSeg{pos + i, (uint64_t)pk, 0u, 0u};
          cur = &seg[n_seg - 1];
        }
        memcpy(pay + pk, raw + i, line_bytes);
        cur->count += (uint32_t)line_bytes;
        pk += line_bytes;
      }
      i += line_bytes;
    }
    pos += i;
    return i;
  }

  size_t finish() {
    if (!ok || pos != total) return 0;
    size_t pay_off = sizeof(PackHeader) + n_seg * sizeof(PackSeg);
    pay_off = (pay_off + 15) & ~(size_t)15;
    const size_t size = pay_off + pk;
    if (size + size / 8 >= total) return 0;   // not worth a second pass on the device
    memmove(out + pay_off, pay, pk);
    PackHeader h{kMagic, (uint32_t)n_seg, total, (uint64_t)pay_off, (uint64_t)pk};
    memcpy(out, &h, sizeof h);
    return size;
  }
};

// The container of raw[0, n) in one go (capacity cap >= pack_bound(n)); 0 = not worth packing.
inline size_t pack(const uint8_t *raw, size_t n, uint8_t *out, size_t cap) {
  Packer p;
  p.begin(out, cap, n);
  p.feed(raw, n, true);
  return p.finish();
}

// Is buf[0, len) a well-formed container?  (Every segment inside the payload, the segments tiling [0, raw_len) in
// order.)  The library checks this before it trusts a table whose offsets index device memory.
inline bool valid(const uint8_t *buf, size_t len) {
  if (len < sizeof(PackHeader)) return false;
  PackHeader h;
  memcpy(&h, buf, sizeof h);
  if (h.magic != kMagic || h.payload_off < sizeof(PackHeader) + (uint64_t)h.n_seg * sizeof(PackSeg) || h.payload_off > len ||
      h.payload_len > len - h.payload_off)
    return false;
  uint64_t raw = 0, pk = 0;
  for (uint32_t k = 0; k < h.n_seg; ++k) {
    PackSeg s;
    memcpy(&s, buf + sizeof(PackHeader) + (size_t)k * sizeof(PackSeg), sizeof s);
    // width is bounded by what the packer writes: (width + 3) / 4 and width + 1 are computed in 32 bits on the device
    // (nq::unpack_kernel), and an unbounded width would let a crafted table index past the wire buffer
    if (s.raw_off != raw || s.pk_off != pk || s.count == 0 || s.width > kMaxWidth) return false;
    const uint64_t rl = seg_raw_len(s), pl = seg_pk_len(s);
    if (pl == 0 || rl > h.raw_len - raw) return false;
    raw += rl;
    pk += pl;
    if (pk > h.payload_len) return false;
  }
  return raw == h.raw_len && pk == h.payload_len;
}

// The file's bytes back (host restatement of nq::unpack_kernel, for tests and for callers without a device).
inline bool unpack(const uint8_t *buf, size_t len, uint8_t *raw_out, size_t cap) {
  if (!valid(buf, len)) return false;
  PackHeader h;
  memcpy(&h, buf, sizeof h);
  if (h.raw_len > cap) return false;
  static const uint8_t canon[4] = {'A', 'C', 'T', 'G'};
  for (uint32_t k = 0; k < h.n_seg; ++k) {
    PackSeg s;
    memcpy(&s, buf + sizeof(PackHeader) + (size_t)k * sizeof(PackSeg), sizeof s);
    const uint8_t *src = buf + h.payload_off + s.pk_off;
    uint8_t *dst = raw_out + s.raw_off;
    if (!s.width) { memcpy(dst, src, s.count); continue; }
    const uint32_t wb = (s.width + 3u) / 4u;
    for (uint32_t l = 0; l < s.count; ++l) {
      for (uint32_t j = 0; j < s.width; ++j) dst[j] = canon[(src[j >> 2] >> (2 * (j & 3u))) & 3u];
      dst[s.width] = '\n';
      dst += s.width + 1;
      src += wb;
    }
  }
  return true;
}

}  // namespace nqp
