"""MI355X-native hot path of the video-classification models (gfx950 HIP kernels behind a C ABI)."""
