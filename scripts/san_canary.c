/* Canary for scripts/sanitize_cpu.sh: two DELIBERATE bugs in a shared object loaded into the same uninstrumented python
 * under the same preloaded runtime as the real libraries.  A clean log only means something if the harness reports
 * these: a heap overflow (ASan) and an unsynchronised counter shared by two threads (TSan). */
#include <pthread.h>
#include <stdlib.h>
static volatile long counter;
static void* bump(void* p) {
    (void)p;
    for (int i = 0; i < 100000; i++) counter++;
    return NULL;
}
long canary_race(void) {
    pthread_t a, b;
    pthread_create(&a, NULL, bump, NULL);
    pthread_create(&b, NULL, bump, NULL);
    pthread_join(a, NULL);
    pthread_join(b, NULL);
    return counter;
}
int canary_overflow(int n) {
    volatile char* p = (volatile char*)malloc(16);
    p[16 + (n & 1)] = 1; /* one past the end */
    int r = p[0];
    free((void*)p);
    return r;
}
