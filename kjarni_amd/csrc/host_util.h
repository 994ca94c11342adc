// Small host-side helpers shared by the loaders and the extern "C" layer.
#pragma once
#include <string>

namespace kjarni {

// Whole file as bytes; throws std::runtime_error("cannot open <path>").
std::string slurp(const std::string& path);
// CStr::to_str would accept it.
bool valid_utf8(const char* s);
// malloc'ed copy for the caller to release with kjarni_string_free; CString::new(..).unwrap_or_default(): a string with
// an interior NUL becomes "".
char* dup_cstr(const std::string& s);

}  // namespace kjarni
