// seqio.h -- FASTA / FASTQ record reader with the reference's exact framing
// (Index::Biogetline, src/niqki_index.cpp:890-941, and get_data_type :944-952).
#pragma once
#include <string>

#include "gzio.h"

namespace nqhost {

// 'Q' when the file name contains ".fq" or ".fastq", else 'A' (:944-952)
inline char data_type(const std::string &filename) {
  if (filename.find(".fq") != std::string::npos) return 'Q';
  if (filename.find(".fastq") != std::string::npos) return 'Q';
  return 'A';
}

// One record: FASTQ = 4 lines; FASTA = header line then every line up to the
// next line starting with '>' (peeked) concatenated.  No upper-casing, no
// trimming.  A result shorter than K is cleared together with the header
// (:912-915).  Returns nothing: callers loop on in.eof() like the reference.
inline void bio_getline(GzReader &in, std::string &result, char type, std::string &header, size_t K) {
  std::string line;
  result.clear();
  if (type == 'Q') {
    in.getline(header);
    in.getline(result);
    in.getline(line);
    in.getline(line);
  } else {
    in.getline(header);
    int c = in.peek();
    while (c != '>' && c != -1) {
      in.getline(line);
      result += line;
      c = in.peek();
    }
  }
  if (result.size() < K) {
    result.clear();
    header.clear();
  }
}

}  // namespace nqhost
