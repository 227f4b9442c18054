// stbdump — TEST INFRASTRUCTURE: runs the image decoder the reference vendors (stb_image v2.19 under
// /root/reference/include/stb_image, compiled from where it lies by the Makefile next to this file) exactly as
// Texture::load calls it (Texture.cpp:40: stbi_load(path, &w, &h, &channels, STBI_rgb_alpha)) and writes
// int32 width, int32 height, then width*height RGBA bytes.  Exit code 2 when stb_image rejects the file.
// Used by make_image_goldens.py to pin cpu-raytracer_amd/host/rtx_image.cpp.
#define STB_IMAGE_IMPLEMENTATION
#include <stb_image/stb_image.h>

#include <stdio.h>

int main(int argc, char ** argv) {
	if (argc != 3) { fprintf(stderr, "usage: stbdump <image> <out.raw>\n"); return 1; }
	int w = 0, h = 0, channels = 0;
	unsigned char * data = stbi_load(argv[1], &w, &h, &channels, STBI_rgb_alpha);
	if (!data) { fprintf(stderr, "stb_image: %s\n", stbi_failure_reason()); return 2; }
	FILE * f = fopen(argv[2], "wb");
	if (!f) return 1;
	fwrite(&w, 4, 1, f); fwrite(&h, 4, 1, f);
	fwrite(data, 4, (size_t)w * h, f);
	fclose(f);
	return 0;
}
