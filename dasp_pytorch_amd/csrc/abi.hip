// ABI stamp of libdasp_hip.so: a hash of include/dasp_hip.h as it was when the library was built (csrc/build.py passes -DDASP_ABI_HASH).
// The torch extension (csrc/torch_ext) is compiled with the same definition and dasp_pytorch_amd._torch_ops refuses to route calls through
// an extension whose hash differs from the library's - a kernel-library-only rebuild after an ABI change must not leave a stale
// libdasp_torch.so calling entry points with yesterday's argument lists.
#ifndef DASP_ABI_HASH
#define DASP_ABI_HASH 0
#endif
extern "C" unsigned long long dasp_abi_hash(void) { return (unsigned long long)DASP_ABI_HASH; }
