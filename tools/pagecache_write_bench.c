/* tools/pagecache_write_bench.c -- how fast can this box take NEW file data into its page cache: T threads, one file each
 * (write() in 8 MiB calls / mmap + memcpy), 4 GiB in total.  What bounds spring_reorder_run's output leg.
 * build: gcc -O2 -pthread -o /tmp/pcw tools/pagecache_write_bench.c ; run: /tmp/pcw DIR */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
static const char *dir;
static size_t per;
static int mode;
static char *src;
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static void *work(void *a) {
  long id = (long)a;
  char p[512];
  snprintf(p, sizeof p, "%s/pcw_%ld.bin", dir, id);
  int fd = open(p, O_RDWR | O_CREAT | O_TRUNC, 0666);
  if (fd < 0) { perror("open"); exit(1); }
  const size_t blk = 8u << 20;
  if (mode == 0) {
    for (size_t o = 0; o < per; o += blk) {
      size_t n = per - o < blk ? per - o : blk, g = 0;
      while (g < n) { ssize_t w = write(fd, src + g, n - g); if (w <= 0) { perror("write"); exit(1); } g += (size_t)w; }
    }
  } else {
    if (ftruncate(fd, (off_t)per)) { perror("ftruncate"); exit(1); }
    char *m = mmap(NULL, per, PROT_READ | PROT_WRITE, MAP_SHARED | (mode == 2 ? MAP_POPULATE : 0), fd, 0);
    if (m == MAP_FAILED) { perror("mmap"); exit(1); }
    for (size_t o = 0; o < per; o += blk) memcpy(m + o, src, per - o < blk ? per - o : blk);
    munmap(m, per);
  }
  close(fd);
  return NULL;
}
int main(int argc, char **argv) {
  dir = argc > 1 ? argv[1] : "/tmp";
  const size_t total = (size_t)4 << 30;
  src = malloc(8u << 20);
  memset(src, 7, 8u << 20);
  for (mode = 0; mode < 3; mode++)
    for (int T = 1; T <= 64; T *= 2) {
      per = total / (size_t)T;
      pthread_t th[64];
      double t0 = now();
      for (long i = 0; i < T; i++) pthread_create(&th[i], NULL, work, (void *)i);
      for (int i = 0; i < T; i++) pthread_join(th[i], NULL);
      double t = now() - t0;
      double t1 = now();
      for (int i = 0; i < T; i++) { char p[512]; snprintf(p, sizeof p, "%s/pcw_%d.bin", dir, i); unlink(p); }
      printf("%s %-14s T=%2d: %6.2f GB/s   (unlink %.3f s)\n", dir, mode == 0 ? "write()" : mode == 1 ? "mmap+memcpy" : "mmap populate", T, total / t / 1e9, now() - t1);
      fflush(stdout);
    }
  return 0;
}
