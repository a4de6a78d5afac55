//! Drop-in body for the reference's src/prover.rs: same signature, the hot path runs on the GPU.
//! Not compiled in the build image (no Rust toolchain); see INTEGRATION.md.
//!
//! `generate_proof` takes the caller's Channel (prover.rs:9) and every commitment and challenge goes
//! through it, exactly as in the reference: the channel's two fields (channel.rs:6-9) cross the FFI
//! with `zk_channel_import`, `zk_prove_channel` runs prover.rs:60-289 on that transcript, and the
//! fields come back for `Channel::finalize` (channel.rs:34-36).  channel.rs needs two crate-private
//! accessors for that (its fields are private); INTEGRATION.md lists the six lines.
use crate::channel::Channel;
use crate::proof::Proof;
use crate::zkstark_amd_sys::*;
use crate::F;
use num_traits::{One, Pow, Zero};

fn check(rc: i32) {
    // the reference panics on any failed check (prover.rs:42-251)
    assert_eq!(rc, ZK_OK, "{:?}", unsafe { std::ffi::CStr::from_ptr(zk_last_error()) });
}

pub fn generate_proof(channel: Channel) -> Proof {
    // prover.rs:32-39 -- the trace stays in Rust (serial recurrence)
    let mut a = [F::zero(); 1023];
    a[0] = F::one();
    a[1] = F::from(3141592);
    for i in 2..1023 {
        a[i] = a[i - 2].pow(2) + a[i - 1].pow(2);
    }
    assert_eq!(a[1022].residue(), 2338775057); // prover.rs:42
    let trace: Vec<u32> = a.iter().map(|f| f.residue()).collect();

    let (state, data) = channel.into_parts(); // pub(crate) accessor added to channel.rs
    unsafe {
        // the library on the path must speak the ABI these declarations were generated from (include/zkstark_amd.h)
        assert_eq!(zk_abi_version(), ZK_ABI_VERSION, "libzkstark_amd.so was built from another zkstark_amd.h");
        let mut ctx = std::ptr::null_mut();
        check(zk_ctx_create(0, 10, 3, &mut ctx)); // n = 1024, blow-up 8 (prover.rs:48-57)
        let mut ch = std::ptr::null_mut();
        check(zk_channel_new(&mut ch));
        check(zk_channel_import(ch, state.as_ptr(), data.as_ptr(), data.len()));
        check(zk_trace_upload(ctx, trace.as_ptr(), trace.len()));
        check(zk_prove_channel(ctx, ch)); // prover.rs:60-289 on the caller's transcript
        let mut out = vec![0u8; zk_channel_data_len(ch)];
        let mut st = [0u8; 32];
        check(zk_channel_data(ch, out.as_mut_ptr(), out.len()));
        check(zk_channel_state(ch, st.as_mut_ptr()));
        zk_channel_free(ch);
        zk_ctx_destroy(ctx);
        Channel::from_parts(st, out).finalize() // prover.rs:292 / channel.rs:34-36
    }
}
