// ModelIO.h — reader / writer of the reference's `.cpb` detector files.
//
// A `.cpb` file is acf::Detector pushed through cereal::PortableBinaryOutputArchive
// (src/lib/acf/io/cereal_pba.h:66-79, load_cpb / save_cpb; Detector::deserializeAny picks this
// path for "*.cpb", ACFIO.cpp:202-231).  cereal itself is a third-party header library that is
// not in this image (Hunter-pinned by the reference), so the PortableBinary rules are restated
// here from its published format, and the field order from the reference's own serialize()
// functions:
//
//   stream      := u8 littleEndianFlag, Detector
//   Detector    := ver, Classifier, Options                        ACFIOArchive.h:75-80 (CEREAL_CLASS_VERSION 1)
//   Classifier  := ver, Mat fids, thrs, child, hs, weights, depth,
//                  vector<f64> errs, losses, i32 treeDepth          ACFIOArchive.h:82-100
//   Mat         := ver, i32 rows, cols, type, u8 continuous, bytes  io/cvmat_cereal.h:20-73
//   Field<T>    := ver, T value, string name, u8 has, u8 isLeaf     ACFField.h:123-130
//   Options ... := the serialize() bodies of ACFIOArchive.h:102-216, in order
//   string      := u64 size, chars;  vector<arithmetic> := u64 size, raw elements
//   ver         := u32 class version, written only the FIRST time a given C++ type is met in the
//                  stream (cereal's per-archive versioned-type set); every type above is versioned
//                  because its serialize() takes a `version` argument.
//
// No sample `.cpb` file ships with the reference, so this layout is "parity unpinned": it is
// checked by an independent Python restatement (acf_amd/modelio.py) producing byte-identical
// files, and by round trips.
#pragma once

#include "HipDetector.h"

#include <iosfwd>
#include <string>

namespace acf
{

// Throws acf::Exception (ACF_HIP_E_INVALID) on a malformed stream.
void loadCpb(std::istream& is, HipDetector::Options& opts, HipDetector::Classifier& clf);
void saveCpb(std::ostream& os, const HipDetector::Options& opts, const HipDetector::Classifier& clf);

// The line-oriented container written by acf_amd/modelio.py (tests, tools).
bool loadAcfm(std::istream& is, HipDetector::Options& opts, HipDetector::Classifier& clf);

// Detector::deserializeAny (ACFIO.cpp:202-231): "*.cpb" -> loadCpb, otherwise the ACFHIPM1 container.
// Returns 0 on success (the reference's convention), non-zero if the file cannot be read.
int loadModelAny(const std::string& filename, HipDetector::Options& opts, HipDetector::Classifier& clf);

} // namespace acf
