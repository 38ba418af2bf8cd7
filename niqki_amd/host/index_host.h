// index_host.h -- host-side mirror of the reference's Index class
// (src/niqki_index.h:35-213) for the file-level drivers: same method names and
// argument meaning, the per-record work (compute_sketch / insert_sketch /
// query_sketch) batched through the C ABI of libniqki_hip.so.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/niqki_hip.h"
#include "gzio.h"

namespace nqhost {

using query_output = std::vector<std::pair<uint32_t, uint32_t>>;  // (count, gid), src/niqki_index.h:31

class Index {
 public:
  // Index(lF,K,W,H,filename,min_fract): src/niqki_index.cpp:13-38
  // n_gpus > 1 (the program's --gpus): the index is cut by sketch-slot range over devices
  // device .. device + n_gpus - 1 (niqki_group_*, RCCL inside libniqki_hip.so); the files of a
  // batch are dealt to the GPUs in order, every GPU frames and sketches its share
  // resident_mib > 0 (the program's --resident-mib): a paged index -- the sketch store in page-locked host
  // memory, the inverted index built for one page of slots at a time within that budget
  Index(uint32_t lF, uint32_t K, uint32_t W, uint32_t H, const std::string &out_filename, double min_fract,
        int device = -1, int n_gpus = 1, int resident_mib = 0);
  // Index(dump file, pretty, filename): src/niqki_index.cpp:63-102
  Index(const std::string &dump_file, bool pretty_printing, const std::string &out_filename, int device = -1,
        int n_gpus = 1, int resident_mib = 0);
  ~Index();
  Index(const Index &) = delete;
  Index &operator=(const Index &) = delete;

  uint32_t K = 0, W = 0, H = 0, lF = 0, F = 0, min_score = 0;
  bool pretty_printing = true;
  std::vector<std::string> filenames;
  std::unique_ptr<ParallelTextWriter> outfile;

  size_t getNbGenomes() const { return filenames.size(); }  // src/niqki_index.h:138-140

  void select_best_H(double genome_size);  // src/niqki_index.cpp:126-138 (-G)

  // per-record operators, kept for API parity (src/niqki_index.h:103,108,142)
  void compute_sketch(const std::string &reference, std::vector<int32_t> &sketch) const;
  void insert_sketch(const std::vector<int32_t> &sketch, uint32_t genome_id);
  query_output query_sketch(const std::vector<int32_t> &sketch) const;

  // file drivers
  void insert_file_of_file_whole(const std::string &filestr);  // :461-500
  void insert_file_lines(const std::string &filestr);          // :383-408
  void query_file_of_file_whole(const std::string &filestr);   // :523-540
  void query_file_lines(const std::string &filestr);           // :412-430
  void query_matrix();                                          // :614-628
  void dump_index_disk(const std::string &filestr);            // :42-59

  void output_query(const query_output &toprint, const std::string &queryname);   // :544-566
  void output_matrix_row(const uint16_t *counts, const std::string &queryname);   // :747-763

 private:
  struct Batch;
  void stage_batch(Batch &b, bool prefetch);
  void flush_insert(Batch &b);
  void flush_query(Batch &b);
  void for_each_batch(const std::vector<std::string> &paths, void (Index::*flush)(Batch &));
  struct Hits {  // hits of a batch of entries: (count, gid) runs hc/hg[off[i] .. off[i+1]) for names[i]
    std::vector<std::string> names;
    std::vector<uint64_t> off;
    std::vector<uint32_t> hc, hg;
    // lines mode: the names are header lines of a piece of the input that stays alive until the WRITER thread has
    // taken them (name e = the line at name_base + name_at[e]); `keep` is that piece
    const uint8_t *name_base = nullptr;
    size_t name_room = 0;
    std::vector<uint64_t> name_at;
    void *keep = nullptr;
  };
  void query_staged(size_t n, Hits &h);
  void write_hits(const Hits &h);
  std::string out_text_;                // write_hits' lines before they go to the writer
  void stream_lines(const std::string &filestr, bool insert);
  void check(int rc, const char *what) const;
  void check_group(int rc, const char *what) const;
  void make_shards(const niqki_params &p, int device, int n_gpus, const uint8_t *dump_header);
  uint32_t per_rank(size_t n) const { return (uint32_t)((n + sh_.size() - 1) / sh_.size()); }
  // multi-GPU: hits of a batch whose entries sit in the shards' staged batches (n_entry[r] of rank r)
  void group_query_staged(uint32_t per, const std::vector<uint32_t> &n_entry, Hits &h);
  niqki_index *h_ = nullptr;            // shard 0 (the only one with one GPU)
  std::vector<niqki_index *> sh_;       // all shards, rank order
  niqki_group *grp_ = nullptr;          // null with one GPU
  // whole-file queries: where flush_query hands a batch's hits (for_each_batch: a writer thread formats and writes them
  // while the next batch is on the GPU); empty: written on the spot
  std::function<void(std::unique_ptr<Hits>)> hits_sink_;
  double t_stage_ = 0, t_dev_ = 0, t_out_ = 0;  // NIQKI_HOST_TIMING: copy + frame / sketch + insert or query / output text
};

}  // namespace nqhost
