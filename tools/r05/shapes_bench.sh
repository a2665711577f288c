#!/bin/bash
# the decoder's main convolution shapes through tools/conv_one.py (N = 120 / 240 at 256^2, 480 at 128^2), one line each
cd "$(dirname "$0")/../.."
for shape in "128 64 3 256 240 p8" "64 32 3 256 240 p8" "99 128 3 256 240" "49 128 3 256 240" "195 128 3 256 120" "128 64 3 128 480 p8" "99 128 3 128 480" "128 128 3 256 16" "96 128 3 256 16" "128 64 3 64 480 p8"; do
  python tools/conv_one.py $shape 2>/dev/null | tail -1
done
