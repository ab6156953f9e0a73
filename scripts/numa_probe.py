"""Diagnostic: NUMA placement of the GPU vs the CPUs this process may run on."""
import glob, os, torch
p = torch.cuda.get_device_properties(0)
bdf = '%04x:%02x:%02x.0' % (getattr(p, 'pci_domain_id', 0), p.pci_bus_id, p.pci_device_id)
print('gpu', p.name, bdf)
for f in ('numa_node', 'local_cpulist'):
    path = '/sys/bus/pci/devices/%s/%s' % (bdf, f)
    print(f, open(path).read().strip() if os.path.exists(path) else 'n/a')
print('affinity', len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8], '...')
print('nodes', [(os.path.basename(n), open(n + '/cpulist').read().strip()) for n in sorted(glob.glob('/sys/devices/system/node/node*'))])
