#include "tga_output.h"

#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

namespace ae {

output::output(const std::string &file_name, u32 width, u32 height, bool tight) {
    const size_t payload = static_cast<size_t>(width) * height * sizeof(u32);
    size_ = tight ? header_size + payload : header_size * payload;

    fd_ = ::open(file_name.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0644);
    if(fd_ == -1) return;
    if(::ftruncate(fd_, static_cast<off_t>(size_)) != 0) return;

    void *map = ::mmap(nullptr, size_, PROT_READ | PROT_WRITE, MAP_SHARED, fd_, 0);
    if(map == MAP_FAILED) return;
    mapping_ = map;

    // 18-byte TGA header (src/output.h:19-34): everything zero except the fields below
    u8 header[header_size] = {};
    header[2] = 2;                                       // image_type: uncompressed true colour
    header[12] = static_cast<u8>(width & 0xff);          // width, u16 little endian
    header[13] = static_cast<u8>((width >> 8) & 0xff);
    header[14] = static_cast<u8>(height & 0xff);         // height
    header[15] = static_cast<u8>((height >> 8) & 0xff);
    header[16] = 32;                                     // pixel depth
    std::memcpy(mapping_, header, header_size);
}

output::~output() {
    if(mapping_) ::munmap(mapping_, size_);
    if(fd_ != -1) ::close(fd_);
}

void *output::get_buffer() const {
    return mapping_ ? static_cast<u8 *>(mapping_) + header_size : nullptr;
}

}
