#include "host_util.h"

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <new>
#include <sstream>
#include <stdexcept>

#include "unicode.h"

namespace kjarni {

std::string slurp(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

bool valid_utf8(const char* s) { return unicode::is_valid_utf8(s, std::strlen(s)); }

char* dup_cstr(const std::string& s)
{
    const bool has_nul = s.find('\0') != std::string::npos;
    const size_t n = has_nul ? 0 : s.size();
    char* p = static_cast<char*>(std::malloc(n + 1));
    if (!p) throw std::bad_alloc();
    std::memcpy(p, s.data(), n);
    p[n] = '\0';
    return p;
}

}  // namespace kjarni
