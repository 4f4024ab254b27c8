for shape in "512 512 256" "1024 1024 64" "2560 2048 8" "5120 5120 1" "3840 2176 4" "1280 704 64"; do
  for q in 1 2; do echo -n "QUAD=$q "; JPEG_AMD_QUAD=$q python tools/run_c3.py 100 $shape 2>/dev/null; done
done
