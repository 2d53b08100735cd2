import sys; sys.path.insert(0,'/root/repo')
import torch, ray_amd
from ray_amd import synth
for k in (31,47,63):
    genome=4_000_000
    n=synth.n_reads_for(genome,150,40)
    p,o,l=synth.make_reads_torch(genome,150,n,3,0.0,device="cuda")
    torch.cuda.synchronize()   # the library runs on its own stream: the generated reads must be complete
    b=ray_amd.GraphBuilder(k, sweep=1)
    b.add_reads_device(p.data_ptr(), n*38, o.data_ptr(), l.data_ptr(), n, 150)
    b.count_kmers()
    st=b.stats()
    print(k, {x:st[x] for x in ("occurrences","candidates","superkmer_records","log_entries","partition_levels")})
    b.close()
