// Baseline JPEG (SOF0, Huffman, 8 bit) to RGB8: what `Image("x.jpg")` textures need
// (src/texture/image.rs:12-20 decodes with the `image` crate; its decoder's IDCT and
// upsampling differ from any other decoder's in the last bit, and so does this one).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace rt {

// Grey or YCbCr (JFIF), any sampling factors up to 4 (half-resolution chroma is interpolated
// with a triangle filter, other ratios replicated), restart intervals.  Progressive, arithmetic-coded, 12-bit and CMYK files are refused.
bool decode_jpeg(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height,
                 std::vector<uint8_t>& rgb, std::string& err);

}  // namespace rt
