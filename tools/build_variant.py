import sys, os, subprocess, shutil
sys.path.insert(0,'/root/repo')
from lightweaver_amd import build
# variants differ in ONE translation unit: copy the base objects, recompile that unit, link
def variant(name, unit, defs):
    base=os.path.join(build.ROOT,'build','liblwhip.so.obj')
    od=os.path.join(build.ROOT,'build','var_'+name+'.obj'); os.makedirs(od,exist_ok=True)
    objs=[]
    for s in build.HIP_SOURCES:
        o=os.path.splitext(s)[0]+'.o'
        if s==unit:
            subprocess.run([build._hipcc(),'--offload-arch=gfx950','-O3','-std=c++17','-fPIC']+['-D'+d for d in defs]+['-c','-o',os.path.join(od,o),os.path.join(build.CSRC,s)],check=True,cwd=build.CSRC)
            objs.append(os.path.join(od,o))
        else:
            objs.append(os.path.join(base,o))
    out=os.path.join(build.PKG,'variants','liblwhip_%s.so'%name)
    subprocess.run([build._hipcc(),'--offload-arch=gfx950','-shared','-fPIC','-o',out]+objs,check=True)
    print(name,'ok')
if __name__=='__main__':
    build.build_hip()
    import json
    for spec in sys.argv[1:]:
        name,unit,defs=spec.split(':')
        variant(name,unit,[d for d in defs.split(',') if d])
