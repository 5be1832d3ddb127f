import hashlib,time,os
d=os.urandom(4*1024*1024+28)
best=9
for _ in range(8):
    t=time.perf_counter(); hashlib.sha512(d).digest(); best=min(best,time.perf_counter()-t)
print("hashlib (OpenSSL) sha512 of 4 MiB: %.2f ms"%(best*1e3))
import subprocess
print(subprocess.run("openssl speed -evp sha512 2>&1 | tail -2; grep -m1 'model name' /proc/cpuinfo; grep -m1 flags /proc/cpuinfo | tr ' ' '\n' | grep -i 'sha\|avx512f\|bmi2' | tr '\n' ' '", shell=True, capture_output=True, text=True).stdout)
