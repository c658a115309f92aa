// Reader for the two Paddle files a reference model directory holds
// (`inference.pdmodel` ProgramDesc protobuf, `inference.pdiparams` LoDTensor records) —
// the inputs of `Config::SetModel(prog, params)` at /root/reference/src/ocr_det.cpp:46.
// Format notes: SURVEY.md §A.4.  No protobuf runtime: a 60-line wire-format walker.
#pragma once
#include <map>
#include <string>
#include <vector>

namespace ocr {

struct HostTensor {
  std::vector<int> dims;
  std::vector<float> data;
  size_t numel() const { return data.size(); }
};

using WeightMap = std::map<std::string, HostTensor>;

// Names of persistable variables of block 0 (feed/fetch excluded), ascending — the order in
// which save_inference_model concatenates them.  Returns false + err on malformed input.
bool pdmodel_persistable_names(const std::string& path, std::vector<std::string>& names, std::string& err);

// Signature of the graph: number of ops of block 0 and FNV-1a 64 over their type names joined by ';' (feed / fetch
// included).  tools/make_plan.py stores the same two numbers in the plan it generates; a model directory whose graph
// differs from the compiled-in plan is refused at load (its weights would be bound to the wrong layers).
bool pdmodel_graph_signature(const std::string& path, int& nops, unsigned long long& fnv, std::string& err);

// Reads every record of a .pdiparams file, pairing them with `names` in order.
bool pdiparams_read(const std::string& path, const std::vector<std::string>& names, WeightMap& out, std::string& err);

bool file_exists(const std::string& path);

}  // namespace ocr
