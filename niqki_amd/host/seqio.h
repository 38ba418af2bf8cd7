// seqio.h -- sequence file type by name, like the reference's get_data_type
// (src/niqki_index.cpp:944-952).  Record framing itself (Index::Biogetline,
// :890-941) runs on the GPU: niqki_stage_raw, niqki_amd/csrc/nq_ingest.hip.
#pragma once
#include <string>

namespace nqhost {

// 'Q' when the file name contains ".fq" or ".fastq", else 'A' (:944-952)
inline char data_type(const std::string &filename) {
  if (filename.find(".fq") != std::string::npos) return 'Q';
  if (filename.find(".fastq") != std::string::npos) return 'Q';
  return 'A';
}

}  // namespace nqhost
