// codegen.h -- the three model-to-text generators behind
// `--model-file X --code-file Y --generator condop|oblivious|vpred`
// (driver.cc:197-224 of the reference).  They read the XML model text as text:
// thresholds and leaf outputs are copied character for character, so the emitted
// scorer computes with exactly the constants the model file holds.
//
//   condop     one C expression of nested `?:` per tree
//              (generate_conditional_operators.cc:28-115)
//   oblivious  table-driven scorer for oblivious ensembles: per tree the features and
//              thresholds of its levels and the 2^depth leaf outputs, trees ordered
//              by depth (generate_oblivious.cc:139-329)
//   vpred      line-oriented breadth-first dump for the VPred scorer
//              (generate_vpred.cc:90-172)
//
// The device scorers (k_score.hip) are the product path; these exist because the
// reference's CLI surface has them.
#pragma once
#include <string>

namespace quickrank {
namespace io {

// each returns the text it would write; `ok` = false when the model could not be
// used (the message is in the returned string)
std::string condop_code(const std::string &model_xml);
std::string oblivious_code(const std::string &model_xml, bool *ok);
std::string vpred_text(const std::string &model_xml, bool *ok);

// reads `model_file`, writes `code_file`; prints the driver's progress line.
// Returns the process exit status (EXIT_SUCCESS / EXIT_FAILURE).
int generate(const std::string &generator, const std::string &model_file,
             const std::string &code_file);

}  // namespace io
}  // namespace quickrank
