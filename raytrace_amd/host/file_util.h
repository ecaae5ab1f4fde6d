// file_util.h -- what the readers ask of a path before they open it.
#ifndef RT_FILE_UTIL_H
#define RT_FILE_UTIL_H

#include <sys/stat.h>

#include <string>

namespace rth {

// A directory opens as a stream without complaint and then fails inside the first read -- with an exception out of the
// stream buffer when the reader pulls characters from the buffer directly.  Every reader therefore takes plain files
// only (a name that came out of a damaged `mtllib`, `path` or `map_Kd` line is often "" or ".").
inline bool is_plain_file(const std::string& path) {
    struct stat st;
    return ::stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

}  // namespace rth
#endif
