// image_codec.h -- image input (texture files of `map_base_color` / `map_subsurface_color`) and output (rgba.png)
// for the pbrlab scene path.  Reference: src/io/image-io.cc:98-224 (which delegates
// to the vendored stb_image / stb_image_write), src/image-utils.cc:8-107 (sRGB transfer functions).
//
// Written from scratch: a zlib inflater/deflater, PNG reader (all colour types and bit depths, Adam7) and writer,
// baseline + progressive JPEG reader, Radiance .hdr reader, scanline OpenEXR reader.  Decoded pixels equal stb_image's / tinyexr's for
// the same file (16-bit PNG samples keep their high byte, sub-byte grey is scaled to 0..255, a tRNS colour key becomes
// an alpha channel; JPEG through stb's integer IDCT, upsampling filters and fixed-point colour conversion).
// BMP, TGA, PNM, GIF and PSD are in image_formats.cpp.  Softimage PIC and tiled or PXR24 / B44 / DWA-compressed OpenEXR are
// NOT decoded by this build: loading such a file fails with a message naming the format.
#ifndef PBRLAB_AMD_IO_IMAGE_CODEC_H_
#define PBRLAB_AMD_IO_IMAGE_CODEC_H_

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace pbio {

// zlib streams (RFC 1950/1951)
bool ZlibInflate(const uint8_t* src, size_t n, std::vector<uint8_t>* out, std::string* err);
void ZlibDeflate(const uint8_t* src, size_t n, std::vector<uint8_t>* out);

// 8-bit PNG, channels 1..4 (grey, grey+alpha, RGB, RGBA)
bool EncodePng(const uint8_t* pixels, size_t width, size_t height, size_t channels, std::vector<uint8_t>* file);
// -> 8 bits per channel, channel count of the file (stbi_load(..., req_comp = 0))
bool DecodePng(const uint8_t* file, size_t n, std::vector<uint8_t>* pixels, size_t* width, size_t* height,
               size_t* channels, std::string* err);
// Radiance RGBE -> 3 floats per pixel (stbi_loadf on a .hdr)
bool DecodeHdr(const uint8_t* file, size_t n, std::vector<float>* pixels, size_t* width, size_t* height,
               std::string* err);

// baseline / extended-sequential / progressive Huffman JPEG (1, 3 or 4 components, any sampling factors, restart intervals)
// -> 1 or 3 channels of 8 bits, the bytes stb_image returns (stbi_load, req_comp = 0).
bool DecodeJpeg(const uint8_t* file, size_t n, std::vector<uint8_t>* pixels, size_t* width, size_t* height, size_t* channels,
                std::string* err);

// The other 8-bit formats stb_image reads (image_formats.cpp): BMP (1/4/8/16/24/32 bits, bit fields; no RLE), TGA (true
// colour, grey, colour-mapped, 15/16-bit, RLE), binary PNM (P5/P6), GIF (first frame, RGBA), PSD (merged RGB image, 8/16
// bits, raw or PackBits).  Is*: the signature / plausibility test stb applies before it tries the format.
bool IsBmp(const uint8_t* file, size_t n);
bool IsGif(const uint8_t* file, size_t n);
bool IsPsd(const uint8_t* file, size_t n);
bool IsPnm(const uint8_t* file, size_t n);
bool IsTga(const uint8_t* file, size_t n);
bool DecodeBmp(const uint8_t* file, size_t n, std::vector<uint8_t>* pixels, size_t* width, size_t* height, size_t* channels, std::string* err);
bool DecodeGif(const uint8_t* file, size_t n, std::vector<uint8_t>* pixels, size_t* width, size_t* height, size_t* channels, std::string* err);
bool DecodePsd(const uint8_t* file, size_t n, std::vector<uint8_t>* pixels, size_t* width, size_t* height, size_t* channels, std::string* err);
bool DecodePnm(const uint8_t* file, size_t n, std::vector<uint8_t>* pixels, size_t* width, size_t* height, size_t* channels, std::string* err);
bool DecodeTga(const uint8_t* file, size_t n, std::vector<uint8_t>* pixels, size_t* width, size_t* height, size_t* channels, std::string* err);
// a Radiance picture opened as an 8-bit image (its name does not end in .hdr): gamma 2.2, clamp, truncate
void HdrToLdr(const std::vector<float>& rgb, std::vector<uint8_t>* out);

// single-part scanline OpenEXR (NONE / RLE / ZIPS / ZIP / PIZ; HALF and FLOAT channels) -> RGBA float as tinyexr's LoadEXR
// returns it (one channel replicated; A = 1 when absent).  Tiled, multipart, PXR24/B44/DWA files are refused.
bool DecodeExr(const uint8_t* file, size_t n, std::vector<float>* pixels, size_t* width, size_t* height, std::string* err);

// io::LoadImageFromFile<float> (image-io.cc:98-152): 8-bit formats are returned as value / 255
bool LoadImageFromFile(const std::string& filename, const std::string& asset_path, std::vector<float>* pixels,
                       size_t* width, size_t* height, size_t* channels);
// io::WritePNG<float> (image-io.cc:172-224): only names ending in ".png"; byte = clamp(x * 256, 0, 255)
bool WritePNG(const std::string& filename, const std::string& asset_path, const std::vector<float>& pixels,
              size_t width, size_t height, size_t channels);
bool WritePNG(const std::string& filename, const std::string& asset_path, const std::vector<uint8_t>& pixels,
              size_t width, size_t height, size_t channels);

// image-utils.cc:8-41 (float instantiations: std::pow(float, float))
float SrgbToLiner(float c_srgb);
float LinerTosRGB(float c_liner);
// image-utils.cc:43-107: the first three channels are converted, a fourth is copied
void SrgbToLiner(const std::vector<float>& src, size_t width, size_t height, size_t channels, std::vector<float>* out);
void LinerToSrgb(const std::vector<float>& src, size_t width, size_t height, size_t channels, std::vector<float>* out);

// pbrlab-cli.cc:47-57: rgba / count -> sRGB -> 8-bit RGBA
void ResolveLayerToSrgb8(const float* rgba, const uint32_t* count, size_t width, size_t height, std::vector<uint8_t>* out);

}  // namespace pbio
#endif
