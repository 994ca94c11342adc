// kjarni-ffi/src/error.rs:7-100 and the runtime trio of lib.rs:35-57.
#include <string>

#include "ffi_common.h"

namespace kjarni {

namespace {
thread_local std::string g_last_error;
thread_local bool g_has_error = false;
}  // namespace

void set_last_error(const std::string& msg)
{
    // CString::new(msg).ok(): an interior NUL yields "no message" in the reference.
    if (msg.find('\0') != std::string::npos) {
        g_has_error = false;
        g_last_error.clear();
        return;
    }
    g_last_error = msg;
    g_has_error = true;
}

}  // namespace kjarni

KJARNI_EXPORT const char* kjarni_error_name(KjarniErrorCode err)
{
    switch (err) {
    case KJARNI_OK: return "KJARNI_OK";
    case KJARNI_ERROR_NULL_POINTER: return "KJARNI_ERROR_NULL_POINTER";
    case KJARNI_ERROR_INVALID_UTF8: return "KJARNI_ERROR_INVALID_UTF8";
    case KJARNI_ERROR_MODEL_NOT_FOUND: return "KJARNI_ERROR_MODEL_NOT_FOUND";
    case KJARNI_ERROR_LOAD_FAILED: return "KJARNI_ERROR_LOAD_FAILED";
    case KJARNI_ERROR_INFERENCE_FAILED: return "KJARNI_ERROR_INFERENCE_FAILED";
    case KJARNI_ERROR_GPU_UNAVAILABLE: return "KJARNI_ERROR_GPU_UNAVAILABLE";
    case KJARNI_ERROR_INVALID_CONFIG: return "KJARNI_ERROR_INVALID_CONFIG";
    case KJARNI_ERROR_CANCELLED: return "KJARNI_ERROR_CANCELLED";
    case KJARNI_ERROR_TIMEOUT: return "KJARNI_ERROR_TIMEOUT";
    case KJARNI_ERROR_STREAM_ENDED: return "KJARNI_ERROR_STREAM_ENDED";
    case KJARNI_ERROR_UNKNOWN: return "KJARNI_ERROR_UNKNOWN";
    }
    return "KJARNI_ERROR_UNKNOWN";
}

KJARNI_EXPORT const char* kjarni_error_code_to_string(KjarniErrorCode err) { return kjarni_error_name(err); }

KJARNI_EXPORT const char* kjarni_last_error_message(void)
{
    return kjarni::g_has_error ? kjarni::g_last_error.c_str() : nullptr;
}

KJARNI_EXPORT void kjarni_clear_error(void)
{
    kjarni::g_has_error = false;
    kjarni::g_last_error.clear();
}

// lib.rs:35-45 sizes the rayon pool; there is no host thread pool to size here.
KJARNI_EXPORT KjarniErrorCode kjarni_init(void) { return KJARNI_OK; }
KJARNI_EXPORT void kjarni_shutdown(void) {}
KJARNI_EXPORT const char* kjarni_version(void) { return "0.1.0"; }
