// nq_api_stage.hip -- raw file bytes in (Index::Biogetline and the read loops around it, src/niqki_index.cpp:383-456,
// :505-519, :890-941): niqki_stage_raw / _prefetch (copy, packed FASTA back to the files' bytes, framing on the device),
// the staged batch (sketch / insert / query / read-back), and the host-side packer's entry points.
#include "nq_handle.h"
#include "nq_pack.h"

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace nqi {

// sketches of the staged entries into their own buffer (once per staged batch; the other
// entry points keep using ws_sk, so they cannot disturb a staged batch)
int staged_sketch_ws(niqki_index *ix) {
  if (!ix->staged.valid) return fail(ix, NIQKI_E_STATE, "no staged batch (niqki_stage_raw first)");
  if (ix->staged.sketched) return NIQKI_OK;
  const uint32_t n = ix->staged.n_entry;
  int rc = ensure(ix, ix->ws_stsk, std::max<size_t>((size_t)n * ix->d.F * 4, 4));
  if (rc) return rc;
  rc = sketch_dev(ix, (const uint8_t *)ix->ws_seq.p, (const uint64_t *)ix->ws_recoff.p, ix->staged.n_rec,
                  ix->staged.entry_rec, n, (int32_t *)ix->ws_stsk.p, ix->staged.entry_bytes);
  if (rc) return rc;
  ix->staged.sketched = true;
  return NIQKI_OK;
}

// A gzip file whose members all say how long they are -- BGZF (the 'B' 'C' subfield of bgzip / htslib: blocks of at most
// 64 KB) or this project's own 'N' 'Q' tag (niqki_amd/host/gzio.h) -- can be cut into its members without inflating
// anything: every member becomes a job of its own, so such a file is inflated by as many wavefronts as it has members.
// members: {offset, bytes, ISIZE} of each; false: not such a file (or not one from end to end).
static bool tagged_members(const uint8_t *c, uint64_t len, std::vector<std::array<uint64_t, 3>> &members) {
  members.clear();
  uint64_t at = 0;
  while (at < len) {
    if (len - at < 28 || c[at] != 0x1F || c[at + 1] != 0x8B || c[at + 2] != 8 || !(c[at + 3] & 4)) return false;
    const uint64_t xlen = (uint64_t)c[at + 10] | (uint64_t)c[at + 11] << 8;
    if (at + 12 + xlen + 8 > len) return false;
    uint64_t total = 0;
    for (uint64_t x = at + 12, xe = at + 12 + xlen; x + 4 <= xe;) {
      const uint64_t sl = (uint64_t)c[x + 2] | (uint64_t)c[x + 3] << 8;
      if (x + 4 + sl > xe) return false;
      if (c[x] == 'B' && c[x + 1] == 'C' && sl == 2) total = ((uint64_t)c[x + 4] | (uint64_t)c[x + 5] << 8) + 1;
      if (c[x] == 'N' && c[x + 1] == 'Q' && sl == 4) total = (uint64_t)c[x + 4] | (uint64_t)c[x + 5] << 8 | (uint64_t)c[x + 6] << 16 | (uint64_t)c[x + 7] << 24;
      x += 4 + sl;
    }
    if (total < 12 + xlen + 8 + 2 || at + total > len) return false;
    const uint8_t *t = c + at + total - 4;
    members.push_back({at, total, (uint64_t)t[0] | (uint64_t)t[1] << 8 | (uint64_t)t[2] << 16 | (uint64_t)t[3] << 24});
    at += total;
  }
  return !members.empty();
}

// The gzip files of a batch through the device inflate (nq_inflate.hip): job j reads d_wire[src, src + src_len) and
// writes d_raw[dst, dst + cap).  Asynchronous on ix->stream; the results (nq::InflateOut per job) lie in ix->ws_ijob
// behind the jobs.
int inflate_launch(niqki_index *ix, const std::vector<nq::InflateJob> &jobs, const uint8_t *d_wire, uint64_t wire_bytes,
                   uint8_t *d_raw) {
  if (jobs.empty()) return NIQKI_OK;
  int rc;
  if (!ix->xtab_ok) {
    if ((rc = ensure(ix, ix->ws_xtab, nq::kInflateXtabWords * 4))) return rc;
    uint32_t t[nq::kInflateXtabWords];
    nq::inflate_xtab(t);
    NQ_HIP(ix, hipMemcpyAsync(ix->ws_xtab.p, t, sizeof t, hipMemcpyHostToDevice, ix->stream));
    NQ_HIP(ix, hipStreamSynchronize(ix->stream));   // (t is a local)
    ix->xtab_ok = true;
  }
  const size_t n = jobs.size();
  if ((rc = ensure(ix, ix->ws_ijob, n * (sizeof(nq::InflateJob) + sizeof(nq::InflateOut))))) return rc;
  NQ_HIP(ix, hipMemcpyAsync(ix->ws_ijob.p, jobs.data(), n * sizeof(nq::InflateJob), hipMemcpyHostToDevice, ix->stream));
  Span sp(ix, NIQKI_KC_INFLATE);
  NQ_HIP(ix, nq::launch_inflate((const nq::InflateJob *)ix->ws_ijob.p, (uint32_t)n, d_wire, wire_bytes, d_raw,
                                (const uint32_t *)ix->ws_xtab.p, (nq::InflateOut *)((nq::InflateJob *)ix->ws_ijob.p + n), ix->stream,
                                ix->inflate_window));
  return NIQKI_OK;
}

}  // namespace nqi

using namespace nqi;

extern "C" {

int niqki_stage_raw_prefetch(niqki_index *ix, const niqki_raw_batch *b) {
  if (!ix || !b) return NIQKI_E_INVALID;
  if (!b->file_ptr || !b->file_off) return fail(ix, NIQKI_E_INVALID, "a prefetch takes the file_ptr form of a host batch");
  NQ_HIP(ix, hipSetDevice(ix->device));
  if (!ix->copy_stream) NQ_HIP(ix, hipStreamCreateWithFlags(&ix->copy_stream, hipStreamNonBlocking));
  const uint32_t slot = ix->pre_next;
  auto &pr = ix->pre[slot];
  if (!pr.ev) NQ_HIP(ix, hipEventCreateWithFlags(&pr.ev, hipEventDisableTiming));
  if (pr.valid) NQ_HIP(ix, hipStreamSynchronize(ix->copy_stream));  // an unused one: its host bytes may go away now
  pr.valid = false;
  const uint32_t nf = b->n_files;
  if (nf == 0) return NIQKI_OK;
  for (uint32_t f = 0; f < nf; ++f)
    if (b->file_off[f + 1] < b->file_off[f]) return fail(ix, NIQKI_E_INVALID, "file_off must be non-decreasing");
  const uint64_t T = b->file_off[nf];
  int rc = ensure(ix, ix->ws_wire[slot], (size_t)T + 2 * NIQKI_SEQ_PAD);
  if (rc) return rc;
  for (uint32_t f = 0; f < nf; ++f) {
    const uint64_t n = b->file_off[f + 1] - b->file_off[f];
    if (n) NQ_HIP(ix, hipMemcpyAsync((uint8_t *)ix->ws_wire[slot].p + b->file_off[f], b->file_ptr[f], n, hipMemcpyHostToDevice, ix->copy_stream));
  }
  NQ_HIP(ix, hipEventRecord(pr.ev, ix->copy_stream));
  pr.ptr.assign(b->file_ptr, b->file_ptr + nf);
  pr.off.assign(b->file_off, b->file_off + nf + 1);
  pr.valid = true;
  ix->pre_next = slot ^ 1u;
  return NIQKI_OK;
}

int niqki_stage_raw(niqki_index *ix, const niqki_raw_batch *b, int mem, niqki_stage_info *info,
                    uint64_t *entry_hdr) {
  if (!ix || !b || !info) return NIQKI_E_INVALID;
  if (b->n_files && (!b->file_off || !b->file_type)) return NIQKI_E_INVALID;
  if (b->lines && b->n_files > 1) return fail(ix, NIQKI_E_INVALID, "lines mode frames one file per call");
  if (b->lines && b->max_entries == 0) return fail(ix, NIQKI_E_INVALID, "max_entries must be > 0");
  NQ_HIP(ix, hipSetDevice(ix->device));
  ix->staged.valid = false;
  ix->staged.sketched = false;
  *info = niqki_stage_info{0, 0, 0, 0};
  const uint32_t nf = b->n_files;
  const uint64_t T = nf ? b->file_off[nf] : 0;   // bytes handed over ("wire" bytes: packed files count as their containers)
  if (nf && !b->raw && !b->file_ptr && T) return NIQKI_E_INVALID;
  if (b->file_ptr && mem != NIQKI_MEM_HOST) return fail(ix, NIQKI_E_INVALID, "file_ptr needs the host memory space");
  // Packed FASTA files (file_type 'a': a container of niqki_pack_fasta): the device writes the file's own bytes back
  // first (nq::unpack_kernel), so everything from here on sees raw files at their raw offsets.
  // Gzip files (NIQKI_FILE_GZIP): the device inflates them (nq::inflate_kernel) to the size their trailers announce.
  bool any_packed = false, any_gz = false;
  for (uint32_t f = 0; f < nf; ++f) {
    any_packed |= b->file_type[f] == 'a';
    any_gz |= (b->file_type[f] & NIQKI_FILE_GZIP) != 0;
  }
  const bool any_wire = any_packed || any_gz;   // some files' bytes are not what crosses PCIe
  if (any_wire && (mem != NIQKI_MEM_HOST || !b->file_ptr || b->lines))
    return fail(ix, NIQKI_E_INVALID, "packed (type 'a') and gzip (NIQKI_FILE_GZIP) files: host memory, the file_ptr form, whole-file mode");
  std::vector<nq::InflateJob> jobs;
  std::vector<uint32_t> job_file;
  std::vector<std::array<uint64_t, 3>> members;
  bool gz_refused = false;
  if (any_gz && b->file_status) std::memset(b->file_status, 0, nf);
  std::vector<uint64_t> roff((size_t)nf + 1, 0);   // raw offsets of the files
  std::vector<nq::UnpackSeg> segs;
  uint64_t unpack_blocks = 0;
  for (uint32_t f = 0; f < nf; ++f) {
    if (b->file_off[f + 1] < b->file_off[f]) return fail(ix, NIQKI_E_INVALID, "file_off must be non-decreasing");
    const uint64_t wire_len = b->file_off[f + 1] - b->file_off[f];
    uint64_t raw_len = wire_len;
    if (b->file_type[f] == 'a') {
      const uint8_t *c = b->file_ptr[f];
      if (!c || !nqp::valid(c, wire_len)) return fail(ix, NIQKI_E_INVALID, "file " + std::to_string(f) + " is not a well-formed packed container");
      nqp::PackHeader h;
      std::memcpy(&h, c, sizeof h);
      raw_len = h.raw_len;
      for (uint32_t k = 0; k < h.n_seg; ++k) {
        nqp::PackSeg ps;
        std::memcpy(&ps, c + sizeof(nqp::PackHeader) + (size_t)k * sizeof(nqp::PackSeg), sizeof ps);
        segs.push_back(nq::UnpackSeg{roff[f] + ps.raw_off, b->file_off[f] + h.payload_off + ps.pk_off, ps.count, ps.width, (uint32_t)unpack_blocks, 0u});
        unpack_blocks += (nqp::seg_raw_len(ps) + nq::kUnpackChunk - 1) / nq::kUnpackChunk;
      }
    } else if (b->file_type[f] & NIQKI_FILE_GZIP) {
      // the size the file announces: its last four bytes (ISIZE of the last member, RFC 1952)
      const uint8_t *c = b->file_ptr[f];
      uint64_t isize = 0;
      if (c && wire_len >= 18) isize = (uint64_t)c[wire_len - 4] | (uint64_t)c[wire_len - 3] << 8 | (uint64_t)c[wire_len - 2] << 16 | (uint64_t)c[wire_len - 1] << 24;
      // a file of size-tagged members (BGZF ...): one job per member
      if (c && wire_len <= 0x7FFF0000ull && tagged_members(c, wire_len, members)) {
        uint64_t sum = 0;
        for (const auto &m : members) sum += m[2];
        if (sum <= 0x7FFF0000ull && sum <= wire_len * 64u) {
          uint64_t at = roff[f];
          for (const auto &m : members) {
            jobs.push_back(nq::InflateJob{b->file_off[f] + m[0], m[1], at, m[2]});
            job_file.push_back(f);
            at += m[2];
          }
          raw_len = sum;
          roff[f + 1] = roff[f] + raw_len;
          continue;
        }
      }
      // not plausibly one plain member (DEFLATE cannot exceed 1032 : 1; FASTA and FASTQ stay below 10 : 1): the host's turn
      if (!c || wire_len < 18 || wire_len > 0x7FFF0000ull || isize > 0x7FFF0000ull || isize > wire_len * 64u || isize * 4096u < wire_len) {
        if (b->file_status) b->file_status[f] = 13;
        gz_refused = true;
        isize = 0;
      } else {
        jobs.push_back(nq::InflateJob{b->file_off[f], wire_len, roff[f], isize});
        job_file.push_back(f);
      }
      raw_len = isize;
    } else if (any_wire && wire_len) {   // a raw file in a batch with packed or gzip'd ones: one raw segment
      if (wire_len > 0xFFFFFFFFull) return fail(ix, NIQKI_E_INVALID, "a raw file of 4 GiB or more cannot share a batch with packed files");
      segs.push_back(nq::UnpackSeg{roff[f], b->file_off[f], (uint32_t)wire_len, 0u, (uint32_t)unpack_blocks, 0u});
      unpack_blocks += (wire_len + nq::kUnpackChunk - 1) / nq::kUnpackChunk;
    }
    roff[f + 1] = roff[f] + raw_len;
  }
  if (unpack_blocks > 0x7FFFFFFFull) return fail(ix, NIQKI_E_INVALID, "raw batch too large");
  if (gz_refused) {
    // The caller reads the refused files again (into the same buffers, as a rule) and stages the batch once more.  If
    // this batch's bytes are on their way from a niqki_stage_raw_prefetch, that copy must have finished with the
    // buffers first; only its slot is given up, a later batch's prefetch stays on its way.
    for (int sl = 0; sl < 2; ++sl) {
      auto &pr = ix->pre[sl];
      if (pr.valid && mem == NIQKI_MEM_HOST && b->file_ptr && nf == pr.ptr.size() && std::equal(pr.ptr.begin(), pr.ptr.end(), b->file_ptr) &&
          std::equal(pr.off.begin(), pr.off.end(), b->file_off)) {
        NQ_HIP(ix, hipEventSynchronize(pr.ev));
        pr.valid = false;
      }
    }
    return fail(ix, NIQKI_E_GZIP, "gzip files whose trailers do not announce a plausible size (file_status)");
  }
  const uint64_t T_raw = roff[nf];
  // chunk table: chunks never span two files
  std::vector<uint8_t> meta((size_t)(nf + 1) * 12 + nf + 16);
  uint64_t *h_off = (uint64_t *)meta.data();
  uint32_t *h_first = (uint32_t *)(meta.data() + (size_t)(nf + 1) * 8);
  uint8_t *h_type = meta.data() + (size_t)(nf + 1) * 12;
  uint64_t chunks = 0;
  for (uint32_t f = 0; f < nf; ++f) {
    const uint8_t ty = b->file_type[f] == 'a' ? (uint8_t)'A' : (uint8_t)(b->file_type[f] & ~NIQKI_FILE_GZIP);
    if (ty != 'A' && ty != 'Q') return fail(ix, NIQKI_E_INVALID, "file_type must be 'A', 'Q', 'a' (packed FASTA) or one of the first two | NIQKI_FILE_GZIP");
    h_off[f] = roff[f];
    h_first[f] = (uint32_t)chunks;
    h_type[f] = ty;
    chunks += (roff[f + 1] - roff[f] + nq::kIngestChunk - 1) / nq::kIngestChunk;
  }
  if (chunks > 0x7FFFFFFFull) return fail(ix, NIQKI_E_INVALID, "raw batch too large");
  h_off[nf] = T_raw;
  h_first[nf] = (uint32_t)chunks;
  int rc;
  const uint8_t *d_raw = b->raw;
  bool prefetched = false;
  // bytes a niqki_stage_raw_prefetch put on their way: this batch's (the other slot's, a later batch's, stay on their
  // way) -- or, when the batch is none of them, all dropped
  uint32_t wslot = 0;   // the wire buffer of this batch
  {
    int hit = -1;
    for (int sl = 0; sl < 2; ++sl) {
      const auto &pr = ix->pre[sl];
      if (pr.valid && mem == NIQKI_MEM_HOST && b->file_ptr && nf == pr.ptr.size() && std::equal(pr.ptr.begin(), pr.ptr.end(), b->file_ptr) &&
          std::equal(pr.off.begin(), pr.off.end(), b->file_off))
        hit = sl;
    }
    if (hit >= 0) {
      prefetched = true;
      wslot = (uint32_t)hit;
      ix->pre[hit].valid = false;
      if (!any_wire) std::swap(ix->ws_raw, ix->ws_wire[wslot]);   // (packed / gzip: the slot stays the wire buffer, unpacked below)
      NQ_HIP(ix, hipStreamWaitEvent(ix->stream, ix->pre[hit].ev, 0));
      d_raw = (const uint8_t *)ix->ws_raw.p;
    } else if (ix->pre[0].valid || ix->pre[1].valid) {
      NQ_HIP(ix, hipStreamSynchronize(ix->copy_stream));
      ix->pre[0].valid = ix->pre[1].valid = false;
      ix->pre_next = 0;
    }
    if (hit < 0) wslot = ix->pre[0].valid ? 1u : 0u;
  }
  if (any_wire) {
    if (!prefetched) {   // the containers (and raw files) as they are, into the wire buffer
      if ((rc = ensure(ix, ix->ws_wire[wslot], (size_t)T + 2 * NIQKI_SEQ_PAD))) return rc;
      for (uint32_t f = 0; f < nf; ++f) {
        const uint64_t n = b->file_off[f + 1] - b->file_off[f];
        if (n) NQ_HIP(ix, hipMemcpyAsync((uint8_t *)ix->ws_wire[wslot].p + b->file_off[f], b->file_ptr[f], n, hipMemcpyHostToDevice, ix->stream));
      }
    }
    if ((rc = ensure(ix, ix->ws_raw, (size_t)T_raw + 2 * NIQKI_SEQ_PAD))) return rc;
    if ((rc = ensure(ix, ix->ws_useg, std::max<size_t>(segs.size() * sizeof(nq::UnpackSeg), 32)))) return rc;
    if (!segs.empty()) {
      NQ_HIP(ix, hipMemcpyAsync(ix->ws_useg.p, segs.data(), segs.size() * sizeof(nq::UnpackSeg), hipMemcpyHostToDevice, ix->stream));
      Span sp(ix, NIQKI_KC_INGEST);
      NQ_HIP(ix, nq::launch_unpack((const nq::UnpackSeg *)ix->ws_useg.p, (uint32_t)segs.size(), (uint32_t)unpack_blocks,
                                   (const uint8_t *)ix->ws_wire[wslot].p, (uint8_t *)ix->ws_raw.p, ix->stream));
    }
    if ((rc = inflate_launch(ix, jobs, (const uint8_t *)ix->ws_wire[wslot].p, (uint64_t)ix->ws_wire[wslot].n & ~(uint64_t)3, (uint8_t *)ix->ws_raw.p))) return rc;
    d_raw = (const uint8_t *)ix->ws_raw.p;
  } else if (prefetched) {
  } else if (mem == NIQKI_MEM_HOST) {
    if ((rc = ensure(ix, ix->ws_raw, (size_t)T + 2 * NIQKI_SEQ_PAD))) return rc;
    if (b->file_ptr) {
      for (uint32_t f = 0; f < nf; ++f) {
        const uint64_t n = b->file_off[f + 1] - b->file_off[f];
        if (n) NQ_HIP(ix, hipMemcpyAsync((uint8_t *)ix->ws_raw.p + b->file_off[f], b->file_ptr[f], n, hipMemcpyHostToDevice, ix->stream));
      }
    } else if (T) {
      NQ_HIP(ix, hipMemcpyAsync(ix->ws_raw.p, b->raw, T, hipMemcpyHostToDevice, ix->stream));
    }
    d_raw = (const uint8_t *)ix->ws_raw.p;
  } else if ((uintptr_t)d_raw & 3) {
    return fail(ix, NIQKI_E_INVALID, "device raw bytes must be 4-byte aligned");
  }
  if ((rc = ensure(ix, ix->ws_fmeta, meta.size()))) return rc;
  if ((rc = ensure(ix, ix->ws_summ, std::max<size_t>((size_t)chunks * 20, 4)))) return rc;
  if ((rc = ensure(ix, ix->ws_chunk, std::max<size_t>((size_t)chunks * 16, 4)))) return rc;
  if ((rc = ensure(ix, ix->ws_fkept, (size_t)(nf + 1) * 8))) return rc;
  if ((rc = ensure(ix, ix->ws_fnrec, (size_t)(nf + 1) * 4))) return rc;
  if ((rc = ensure(ix, ix->ws_misc, 256))) return rc;
  NQ_HIP(ix, hipMemcpyAsync(ix->ws_fmeta.p, meta.data(), meta.size(), hipMemcpyHostToDevice, ix->stream));
  nq::IngestArgs a;
  a.raw = d_raw;
  a.file_off = (const uint64_t *)ix->ws_fmeta.p;
  a.chunk_first = (const uint32_t *)((uint8_t *)ix->ws_fmeta.p + (size_t)(nf + 1) * 8);
  a.file_type = (const uint8_t *)ix->ws_fmeta.p + (size_t)(nf + 1) * 12;
  a.n_files = nf;
  a.n_chunks = (uint32_t)chunks;
  a.summ = (uint32_t *)ix->ws_summ.p;
  a.chunk_out = (uint32_t *)ix->ws_chunk.p;
  a.file_kept = (uint64_t *)ix->ws_fkept.p;
  a.file_nrec = (uint32_t *)ix->ws_fnrec.p;
  a.totals = (uint64_t *)ix->ws_misc.p;
  a.seqs = nullptr;
  a.rec_off = nullptr;
  a.hdr_pos = nullptr;
  uint64_t totals[2] = {0, 0};
  {
    Span sp(ix, NIQKI_KC_INGEST);
    NQ_HIP(ix, nq::launch_ingest_scan(a, ix->stream));
  }
  NQ_HIP(ix, hipMemcpyAsync(totals, a.totals, 16, hipMemcpyDeviceToHost, ix->stream));
  std::vector<nq::InflateOut> iout(jobs.size());
  if (!jobs.empty())
    NQ_HIP(ix, hipMemcpyAsync(iout.data(), (const nq::InflateJob *)ix->ws_ijob.p + jobs.size(), jobs.size() * sizeof(nq::InflateOut),
                              hipMemcpyDeviceToHost, ix->stream));
  NQ_HIP(ix, hipStreamSynchronize(ix->stream));  // also: `meta` and the caller's raw bytes are consumed
  {
    uint32_t bad = 0;
    for (int i = 0; i < 4; ++i) ix->inflate_stats[i] = 0;
    for (size_t j = 0; j < jobs.size(); ++j) {
      ix->inflate_stats[0] += iout[j].rounds; ix->inflate_stats[1] += iout[j].round_bytes;
      ix->inflate_stats[2] += iout[j].serial_tokens; ix->inflate_stats[3] += iout[j].blocks;
    }
    for (size_t j = 0; j < jobs.size(); ++j)
      if (iout[j].status) {
        if (b->file_status) b->file_status[job_file[j]] = (uint8_t)iout[j].status;
        ++bad;
      }
    if (bad) return fail(ix, NIQKI_E_GZIP, std::to_string(bad) + " gzip file(s) not taken by the device inflate (file_status)");
  }
  if (totals[0] > 0xFFFFFFF0ull) return fail(ix, NIQKI_E_INVALID, "too many records in one batch");
  const uint32_t n_rec = (uint32_t)totals[0];
  const uint64_t kept = totals[1];
  if ((rc = ensure(ix, ix->ws_recoff, (size_t)(n_rec + 1) * 8))) return rc;
  if ((rc = ensure(ix, ix->ws_hdrpos, std::max<size_t>((size_t)n_rec * 8, 8)))) return rc;
  if ((rc = ensure(ix, ix->ws_seq, (size_t)kept + 2 * NIQKI_SEQ_PAD))) return rc;
  a.seqs = (uint8_t *)ix->ws_seq.p;
  a.rec_off = (uint64_t *)ix->ws_recoff.p;
  a.hdr_pos = (uint64_t *)ix->ws_hdrpos.p;
  {
    Span sp(ix, NIQKI_KC_INGEST);
    NQ_HIP(ix, nq::launch_ingest_emit(a, ix->stream));
  }
  NQ_HIP(ix, hipMemcpyAsync(a.rec_off + n_rec, a.totals + 1, 8, hipMemcpyDeviceToDevice, ix->stream));
  NQ_HIP(ix, hipMemsetAsync(a.seqs + kept, 0, NIQKI_SEQ_PAD, ix->stream));
  uint32_t n_entry = nf;
  uint64_t consumed = T_raw, entry_bytes = kept;
  const uint32_t *d_entry = a.file_nrec;  // whole mode: entry f = the records of file f
  if (b->lines) {
    const uint32_t n_use = b->final ? n_rec : (n_rec ? n_rec - 1 : 0);
    if ((rc = ensure(ix, ix->ws_entry, (size_t)(b->max_entries + 1) * 4))) return rc;
    if ((rc = ensure(ix, ix->ws_ehdr, (size_t)b->max_entries * 8))) return rc;
    uint32_t *d_res = (uint32_t *)((uint8_t *)ix->ws_misc.p + 64);
    NQ_HIP(ix, nq::launch_ingest_entries(a.rec_off, a.hdr_pos, n_use, ix->d.K, b->max_entries,
                                         (uint32_t *)ix->ws_entry.p, (uint64_t *)ix->ws_ehdr.p, d_res, ix->stream));
    uint32_t res[2] = {0, 0};
    NQ_HIP(ix, hipMemcpyAsync(res, d_res, 8, hipMemcpyDeviceToHost, ix->stream));
    NQ_HIP(ix, hipStreamSynchronize(ix->stream));
    n_entry = res[0];
    if (res[1] < n_rec) {
      NQ_HIP(ix, hipMemcpyAsync(&consumed, a.hdr_pos + res[1], 8, hipMemcpyDeviceToHost, ix->stream));
      // (the launch shape of the sketch kernel goes by the entries' average length: the records behind the last entry
      // are not theirs -- a piece of 50 000 reads staged 16 384 entries at a time looked like records of 457 bases)
      NQ_HIP(ix, hipMemcpyAsync(&entry_bytes, a.rec_off + res[1], 8, hipMemcpyDeviceToHost, ix->stream));
    }
    if (entry_hdr && n_entry)
      NQ_HIP(ix, hipMemcpyAsync(entry_hdr, ix->ws_ehdr.p, (size_t)n_entry * 8, hipMemcpyDeviceToHost, ix->stream));
    NQ_HIP(ix, hipStreamSynchronize(ix->stream));
    d_entry = (const uint32_t *)ix->ws_entry.p;
  }
  ix->staged.valid = true;
  ix->staged.n_entry = n_entry;
  ix->staged.n_rec = n_rec;
  ix->staged.seq_bytes = kept;
  ix->staged.entry_bytes = entry_bytes;
  ix->staged.entry_rec = d_entry;
  info->n_entry = n_entry;
  info->n_rec = n_rec;
  info->consumed = consumed;
  info->seq_bytes = kept;
  return NIQKI_OK;
}

int niqki_staged_sketch(niqki_index *ix, int32_t *sketches, int mem) {
  if (!ix) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  int rc = staged_sketch_ws(ix);
  if (rc) return rc;
  const size_t bytes = (size_t)ix->staged.n_entry * ix->d.F * 4;
  if (!bytes) return NIQKI_OK;
  if (!sketches) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipMemcpyAsync(sketches, ix->ws_stsk.p, bytes,
                            mem == NIQKI_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, ix->stream));
  if (mem == NIQKI_MEM_HOST) NQ_HIP(ix, hipStreamSynchronize(ix->stream));
  return NIQKI_OK;
}

int niqki_staged_insert(niqki_index *ix) {
  if (!ix) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  int rc = staged_sketch_ws(ix);
  if (rc) return rc;
  return niqki_insert(ix, (const int32_t *)ix->ws_stsk.p, ix->staged.n_entry, NIQKI_MEM_DEVICE);
}

int niqki_staged_query(niqki_index *ix, uint64_t *hit_off, uint32_t *hit_counts, uint32_t *hit_gids,
                       uint64_t capacity, int mem) {
  if (!ix || !hit_off) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  int rc = staged_sketch_ws(ix);
  if (rc) return rc;
  if (mem == NIQKI_MEM_DEVICE)
    return niqki_query(ix, (const int32_t *)ix->ws_stsk.p, ix->staged.n_entry, hit_off, hit_counts, hit_gids,
                       capacity, NIQKI_MEM_DEVICE);
  if ((rc = build_if_needed(ix))) return rc;
  return query_to_host(ix, (const int32_t *)ix->ws_stsk.p, true, ix->staged.n_entry, hit_off, hit_counts,
                       hit_gids, capacity);
}

int niqki_staged_records(niqki_index *ix, uint64_t *rec_off, uint8_t *seqs, uint32_t *entry_rec,
                         uint64_t *hdr_pos) {
  if (!ix) return NIQKI_E_INVALID;
  if (!ix->staged.valid) return fail(ix, NIQKI_E_STATE, "no staged batch (niqki_stage_raw first)");
  NQ_HIP(ix, hipSetDevice(ix->device));
  const auto &st = ix->staged;
  if (rec_off) NQ_HIP(ix, hipMemcpyAsync(rec_off, ix->ws_recoff.p, (size_t)(st.n_rec + 1) * 8, hipMemcpyDeviceToHost, ix->stream));
  if (seqs && st.seq_bytes) NQ_HIP(ix, hipMemcpyAsync(seqs, ix->ws_seq.p, st.seq_bytes, hipMemcpyDeviceToHost, ix->stream));
  if (entry_rec) NQ_HIP(ix, hipMemcpyAsync(entry_rec, st.entry_rec, (size_t)(st.n_entry + 1) * 4, hipMemcpyDeviceToHost, ix->stream));
  if (hdr_pos && st.n_rec) NQ_HIP(ix, hipMemcpyAsync(hdr_pos, ix->ws_hdrpos.p, (size_t)st.n_rec * 8, hipMemcpyDeviceToHost, ix->stream));
  NQ_HIP(ix, hipStreamSynchronize(ix->stream));
  return NIQKI_OK;
}

int niqki_gunzip(niqki_index *ix, const uint8_t *gz, const uint64_t *gz_off, uint32_t n_files, const uint64_t *raw_off,
                 uint8_t *raw, uint32_t *status, uint64_t *produced, uint32_t *members, uint64_t *outside) {
  if (!ix || !gz || !gz_off || !raw_off) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  ix->staged.valid = false;   // (shares the staging buffers)
  if (ix->pre[0].valid || ix->pre[1].valid) { NQ_HIP(ix, hipStreamSynchronize(ix->copy_stream)); ix->pre[0].valid = ix->pre[1].valid = false; ix->pre_next = 0; }
  if (n_files == 0) return NIQKI_OK;
  std::vector<nq::InflateJob> jobs(n_files);
  for (uint32_t f = 0; f < n_files; ++f) {
    if (gz_off[f + 1] < gz_off[f] || raw_off[f + 1] < raw_off[f]) return fail(ix, NIQKI_E_INVALID, "offsets must be non-decreasing");
    jobs[f] = nq::InflateJob{gz_off[f], gz_off[f + 1] - gz_off[f], raw_off[f], raw_off[f + 1] - raw_off[f]};
  }
  const uint64_t T = gz_off[n_files], T_raw = raw_off[n_files];
  int rc;
  if ((rc = ensure(ix, ix->ws_wire[0], (size_t)T + 2 * NIQKI_SEQ_PAD))) return rc;
  if ((rc = ensure(ix, ix->ws_raw, (size_t)T_raw + 2 * NIQKI_SEQ_PAD))) return rc;
  if (T > gz_off[0]) NQ_HIP(ix, hipMemcpyAsync((uint8_t *)ix->ws_wire[0].p + gz_off[0], gz + gz_off[0], T - gz_off[0], hipMemcpyHostToDevice, ix->stream));
  NQ_HIP(ix, hipMemsetAsync(ix->ws_raw.p, 0xEE, (size_t)T_raw + 2 * NIQKI_SEQ_PAD, ix->stream));
  if ((rc = inflate_launch(ix, jobs, (const uint8_t *)ix->ws_wire[0].p, (uint64_t)ix->ws_wire[0].n & ~(uint64_t)3, (uint8_t *)ix->ws_raw.p))) return rc;
  std::vector<nq::InflateOut> out(n_files);
  NQ_HIP(ix, hipMemcpyAsync(out.data(), (const nq::InflateJob *)ix->ws_ijob.p + n_files, (size_t)n_files * sizeof(nq::InflateOut),
                            hipMemcpyDeviceToHost, ix->stream));
  std::vector<uint8_t> whole;
  if (outside) {
    whole.resize((size_t)T_raw + 2 * NIQKI_SEQ_PAD);
    NQ_HIP(ix, hipMemcpyAsync(whole.data(), ix->ws_raw.p, whole.size(), hipMemcpyDeviceToHost, ix->stream));
  } else if (raw && T_raw > raw_off[0]) {
    NQ_HIP(ix, hipMemcpyAsync(raw + raw_off[0], (const uint8_t *)ix->ws_raw.p + raw_off[0], T_raw - raw_off[0], hipMemcpyDeviceToHost, ix->stream));
  }
  NQ_HIP(ix, hipStreamSynchronize(ix->stream));
  if (outside) {
    // a file may only have written the first `produced` bytes of its own range
    uint64_t diff = 0, at = 0;
    for (uint32_t f = 0; f <= n_files; ++f) {
      const uint64_t lo = f < n_files ? raw_off[f] : whole.size();
      for (uint64_t i = at; i < lo; ++i) diff += whole[i] != 0xEE;
      if (f < n_files) {
        const uint64_t made = std::min<uint64_t>(out[f].produced, raw_off[f + 1] - raw_off[f]);
        for (uint64_t i = raw_off[f] + made; i < raw_off[f + 1]; ++i) diff += whole[i] != 0xEE;
        at = raw_off[f + 1];
      }
    }
    *outside = diff;
    if (raw && T_raw > raw_off[0]) std::memcpy(raw + raw_off[0], whole.data() + raw_off[0], T_raw - raw_off[0]);
  }
  for (int i = 0; i < 4; ++i) ix->inflate_stats[i] = 0;
#ifdef NQ_INFLATE_CLOCK
  fprintf(stderr, "inflate clocks of file 0 (outside, decode, walk, copy, reader, flush, tables, serial matches):");
  for (int i = 0; i < 8; ++i) fprintf(stderr, " %llu", (unsigned long long)out[0].clk[i]);
  fprintf(stderr, "\n");
#endif
  for (uint32_t f = 0; f < n_files; ++f) {
    ix->inflate_stats[0] += out[f].rounds; ix->inflate_stats[1] += out[f].round_bytes;
    ix->inflate_stats[2] += out[f].serial_tokens; ix->inflate_stats[3] += out[f].blocks;
    if (status) status[f] = out[f].status;
    if (produced) produced[f] = out[f].produced;
    if (members) members[f] = out[f].members;
  }
  return NIQKI_OK;
}

int niqki_gunzip_stats(niqki_index *ix, uint64_t out[4]) {
  if (!ix || !out) return NIQKI_E_INVALID;
  for (int i = 0; i < 4; ++i) out[i] = ix->inflate_stats[i];
  return NIQKI_OK;
}

// ---- packed FASTA (nq_pack.h): host code, no device needed ----
size_t niqki_pack_bound(size_t n) { return nqp::pack_bound(n); }

size_t niqki_pack_fasta(const uint8_t *raw, size_t n, uint8_t *out, size_t capacity) {
  if (!raw || !out) return 0;
  return nqp::pack(raw, n, out, capacity);
}

int niqki_unpack_fasta(const uint8_t *container, size_t len, uint8_t *raw, size_t capacity, size_t *raw_len) {
  if (!container || !nqp::valid(container, len)) return NIQKI_E_INVALID;
  nqp::PackHeader h;
  std::memcpy(&h, container, sizeof h);
  if (raw_len) *raw_len = (size_t)h.raw_len;
  if (!raw) return NIQKI_OK;
  if (h.raw_len > capacity) return NIQKI_E_CAPACITY;
  return nqp::unpack(container, len, raw, capacity) ? NIQKI_OK : NIQKI_E_INVALID;
}

}  // extern "C"
