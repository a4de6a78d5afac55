//! Drop-in body for the reference's src/prover.rs: same signature, the hot path runs on the GPU.
//! Not compiled in the build image (no Rust toolchain); see INTEGRATION.md.
use crate::channel::Channel;
use crate::proof::Proof;
use crate::zkstark_amd_sys::*;
use crate::F;
use num_traits::{One, Pow, Zero};

pub fn generate_proof(_channel: Channel) -> Proof {
    // prover.rs:32-39 -- the trace stays in Rust (serial recurrence)
    let mut a = [F::zero(); 1023];
    a[0] = F::one();
    a[1] = F::from(3141592);
    for i in 2..1023 {
        a[i] = a[i - 2].pow(2) + a[i - 1].pow(2);
    }
    assert_eq!(a[1022].residue(), 2338775057); // prover.rs:42
    let trace: Vec<u32> = a.iter().map(|f| f.residue()).collect();

    unsafe {
        let mut ctx = std::ptr::null_mut();
        assert_eq!(zk_ctx_create(0, 10, 3, &mut ctx), ZK_OK); // n = 1024, blow-up 8 (prover.rs:48-57)
        let cap = zk_proof_data_len(10, 3);
        let mut data = vec![0u8; cap];
        let (mut len, mut state) = (0usize, [0u8; 32]);
        let rc = zk_prove(ctx, trace.as_ptr(), trace.len(), data.as_mut_ptr(), cap, &mut len, state.as_mut_ptr());
        zk_ctx_destroy(ctx);
        // the reference panics on any failed check (prover.rs:42-251)
        assert_eq!(rc, ZK_OK, "{:?}", std::ffi::CStr::from_ptr(zk_last_error()));
        data.truncate(len);
        Proof::new(state, data.into_boxed_slice()) // proof.rs:11; Channel::finalize moves the same two fields (channel.rs:34-36)
    }
}
