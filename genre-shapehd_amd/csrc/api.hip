// api.hip -- library-level entry points of libgenre_hip.so (error string, ABI version).
#include "common.hpp"

namespace genre {
char *err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace genre

extern "C" int genre_abi_version(void) { return GENRE_ABI_VERSION; }
extern "C" const char *genre_last_error(void) { return genre::err_buf(); }
