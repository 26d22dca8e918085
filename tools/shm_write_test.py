import os, time, threading
N = 3 << 30
buf = bytes(64 << 20)
def run(pre, threads=8):
    p = "/dev/shm/_wtest"
    try: os.unlink(p)
    except FileNotFoundError: pass
    fd = os.open(p, os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    t0 = time.time()
    if pre:
        os.posix_fallocate(fd, 0, N)
    t1 = time.time()
    # chunks of 64 MiB, each written by `threads` threads in slices (like the executable)
    for off in range(0, N, len(buf)):
        sl = len(buf) // threads
        th = [threading.Thread(target=lambda k=k: os.pwrite(fd, memoryview(buf)[k*sl:(k+1)*sl], off + k*sl)) for k in range(threads)]
        [t.start() for t in th]; [t.join() for t in th]
    t2 = time.time()
    os.close(fd); os.unlink(p)
    return t1 - t0, t2 - t1
for pre in (False, True, False, True):
    a, b = run(pre)
    print("fallocate" if pre else "plain    ", f"prealloc {a:.2f} s  write {b:.2f} s  -> {N / (a + b) / 1e9:.2f} GB/s total, write alone {N / b / 1e9:.2f} GB/s")
