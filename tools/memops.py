import re,glob,collections,sys,os,subprocess,tempfile,shutil
lib=sys.argv[1]
d=tempfile.mkdtemp()
shutil.copy(lib, d+'/lib.so')
subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump','--offloading','lib.so'],cwd=d,capture_output=True)
for fn in glob.glob(d+'/lib.so.*gfx950'):
    txt=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump','-d',fn],capture_output=True,text=True).stdout
    for f in re.split(r'\n(?=[0-9a-f]+ <)', txt):
        m=re.match(r'[0-9a-f]+ <([^>]+)>', f)
        if not m: continue
        name=m.group(1)
        if not any(k in name for k in sys.argv[2:]): continue
        c=collections.Counter()
        for line in f.split('\n')[1:]:
            mm=re.match(r'\s+(scratch|global|flat)_(load|store|atomic)', line)
            if mm: c[mm.group(1)+'_'+mm.group(2)]+=1
        print(name[:44], dict(c))
shutil.rmtree(d)
