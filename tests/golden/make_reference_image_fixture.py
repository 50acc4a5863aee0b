"""Makes tests/golden/reference_image/: the one output image of the reference itself whose input is also in the
reference tree -- images/eorovan.blend.rts.bmp, saved by the reference's own export key (SDL_SaveBMP of the
accumulated frame, kernel.cu K:2505-2513) for samples/eorovan.blend.rts with that file's own '*' settings line.
Data only: the image is re-encoded losslessly as PNG, the scene text is gzipped.  (The scene's texture
eurovan_dif_red.ppm is not in the reference tree, so the van's colour cannot be compared; camera, sky, silhouette
and the untextured glossy floor can.)   Run in the build container: python tests/golden/make_reference_image_fixture.py"""
import gzip
import os
import shutil

from PIL import Image

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_image")
os.makedirs(OUT, exist_ok=True)
Image.open(os.path.join(REF, "images", "eorovan.blend.rts.bmp")).convert("RGB").save(os.path.join(OUT, "eorovan.blend.rts.png"), optimize=True)
with open(os.path.join(REF, "samples", "eorovan.blend.rts"), "rb") as f, gzip.GzipFile(os.path.join(OUT, "eorovan.blend.rts.gz"), "wb", mtime=0) as g:
    shutil.copyfileobj(f, g)

# Second pair: images/bolter2.blend.rts.bmp for samples/bolter2.blend.rts (albedo texture boltersmall.ppm on the gun,
# env.ppm as environment map).  The reference's window lets the user move the camera with the arrow keys before
# saving (campos.x -+ 1, campos.z -+ 1 per press, K:2354-2376); the saved frame is the scene's '*' camera after
# LEFT x3 and DOWN x2 (found by searching the key-step lattice; nothing else was changed).
Image.open(os.path.join(REF, "images", "bolter2.blend.rts.bmp")).convert("RGB").save(os.path.join(OUT, "bolter2.blend.rts.png"), optimize=True)
with open(os.path.join(REF, "samples", "bolter2.blend.rts"), "rb") as f, gzip.GzipFile(os.path.join(OUT, "bolter2.blend.rts.gz"), "wb", mtime=0) as g:
    shutil.copyfileobj(f, g)
for tex in ("boltersmall", "env"):      # P6 pixel data, re-encoded losslessly (the test writes them back as P6)
    Image.open(os.path.join(REF, "samples", tex + ".ppm")).convert("RGB").save(os.path.join(OUT, tex + ".png"), optimize=True)
# the 122 header bytes of the saved BMP: the export format dogeray_main.cpp reproduces
with open(os.path.join(REF, "images", "eorovan.blend.rts.bmp"), "rb") as f, open(os.path.join(OUT, "eorovan.blend.rts.bmp.header"), "wb") as g:
    g.write(f.read(122))
print({n: os.path.getsize(os.path.join(OUT, n)) for n in sorted(os.listdir(OUT))})
