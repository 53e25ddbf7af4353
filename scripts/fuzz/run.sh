#!/bin/bash
# AddressSanitizer + UBSan mutation fuzz of the standalone readers (CPU only; sanitizers cannot run on the GPU pool).
# usage: scripts/fuzz/run.sh [mutations per seed file, default 300]
set -e
here=$(cd $(dirname $0) && pwd); io=$here/../../pbrlab_amd/csrc/io; gold=$here/../../tests/golden/io
mkdir -p /tmp/pbrio_fuzz
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -I$io -I$here/../../include $here/readers_fuzz.cpp \
    $io/obj_reader.cpp $io/hair_reader.cpp $io/image_codec.cpp $io/image_formats.cpp -o /tmp/pbrio_fuzz/fuzz
printf 'mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl A\nf 1 2 3\n' > /tmp/pbrio_fuzz/host.obj
cd $gold && /tmp/pbrio_fuzz/fuzz ${1:-300} tex0.png tex3.png tex5.png tex8.png tex10.png tex12.png photo0.jpg photo1.jpg photo3.jpg \
    photo4.jpg photo6.jpg photo7.jpg env0.exr env1.exr env2.exr env3.exr env5.exr env6.exr env7.exr env8.exr env0.hdr env1.hdr env2.hdr case0.obj case5.obj \
    case11.obj strands0.hair strands1.hair strands3.hair strands4.hair \
    o_24.bmp o_32_topdown.bmp o_pal4.bmp o_pal1_v4.bmp o_565.bmp o_4444_v5.bmp o_rgb_rle.tga o_rgba_topdown.tga o_555.tga \
    o_grey_alpha_rle.tga o_indexed.tga o_indexed16.tga o.ppm o.pgm o_plain.gif o_interlaced_transparent.gif o_local_palette.gif \
    o_rgb16.psd o_rgba_rle.psd o_5ch.psd o_radiance.pic prog0.jpg prog1.jpg prog2.jpg
# host SAH builder: random / degenerate primitive sets, structural validation of the flattened tree
cs=$here/../../pbrlab_amd/csrc
/opt/rocm/bin/hipcc --offload-host-only -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -I$cs -I$here/../../include \
    $here/bvh_check.cpp $cs/bvh_build.cpp -o /tmp/pbrio_fuzz/bvh_check -lpthread
/tmp/pbrio_fuzz/bvh_check
