//! Safe layer over `mi355rt-sys`.  `Gpu::render` stands where the reference's
//! `rayon::iter::split(..).for_each(..)` block stands (src/main.rs:65-101): a realised scene in,
//! per-pixel sums of `ray_color` over the samples out; `main()` keeps `into_srgb8`, the PNG and
//! the `Done in` line (src/main.rs:98, :103-115).
//!
//! Source only: written without a Rust toolchain at hand.
pub mod flatten;

use anyhow::{anyhow, Context, Result};
use mi355rt_sys as sys;
use std::ffi::CStr;

pub struct Gpu {
    ctx: *mut sys::rt_ctx,
    _devices: Vec<i32>,
    // boxed twice: the library keeps a thin pointer to the inner (fat) box
    progress: Option<Box<Box<dyn FnMut(u64, u64) + Send>>>,
}

unsafe extern "C" fn progress_thunk(user: *mut std::os::raw::c_void, done: u64, total: u64) {
    let f = &mut *(user as *mut Box<dyn FnMut(u64, u64) + Send>);
    f(done, total);
}

impl Gpu {
    /// One device, or a device group (`devices[0]` is the root the shares are reduced onto).
    pub fn new(devices: &[i32]) -> Result<Self> {
        let devices = devices.to_vec();
        let cfg = sys::rt_config {
            device: devices.first().copied().unwrap_or(0),
            n_devices: devices.len() as u32,
            max_paths_in_flight: 0,
            devices: if devices.is_empty() { std::ptr::null() } else { devices.as_ptr() },
        };
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { sys::rt_create(&cfg, &mut ctx) };
        anyhow::ensure!(rc == sys::RT_OK, "no usable HIP device (rt_create: {rc})");
        Ok(Self { ctx, _devices: devices, progress: None })
    }

    fn error(&self) -> anyhow::Error {
        anyhow!(unsafe { CStr::from_ptr(sys::rt_last_error(self.ctx)) }.to_string_lossy().into_owned())
    }

    pub fn upload(&mut self, scene: &flatten::FlatScene) -> Result<()> {
        let view = scene.view();
        if unsafe { sys::rt_upload_scene(self.ctx, &view) } != sys::RT_OK {
            return Err(self.error()).context("uploading scene");
        }
        Ok(())
    }

    /// `f(done, total)` in pixel-samples after every finished pass of a render: what drives the
    /// `indicatif` bar of src/main.rs:58-61 (`bar.set_position(done / samples_per_pixel)`).
    /// A device group calls it from its member threads, one at a time, hence `Send`.
    pub fn set_progress(&mut self, f: Option<Box<dyn FnMut(u64, u64) + Send>>) {
        self.progress = f.map(Box::new);
        match self.progress.as_mut() {
            Some(b) => unsafe {
                sys::rt_set_progress(self.ctx, Some(progress_thunk), &mut **b as *mut Box<_> as *mut _);
            },
            None => unsafe {
                sys::rt_set_progress(self.ctx, None, std::ptr::null_mut());
            },
        }
    }

    /// Sums over samples `[0, samples_per_pixel)` of `ray_color` per pixel, row 0 on top, RGB.
    pub fn render(&mut self, width: u32, height: u32, samples_per_pixel: u32, max_depth: u32, seed: u64) -> Result<Vec<f64>> {
        let p = sys::rt_render_params {
            width, height, sample_begin: 0, sample_end: samples_per_pixel, max_depth, flags: 0, seed,
            window: sys::rt_window::default(),
        };
        let n = width as usize * height as usize * 3;
        if self._devices.len() > 1 {
            // a device group adds float shares
            let mut sum = vec![0f32; n];
            if unsafe { sys::rt_render(self.ctx, &p, sum.as_mut_ptr(), std::ptr::null_mut()) } != sys::RT_OK {
                return Err(self.error()).context("rendering");
            }
            return Ok(sum.into_iter().map(f64::from).collect());
        }
        let mut sum = vec![0f64; n];
        if unsafe { sys::rt_render_f64(self.ctx, &p, sum.as_mut_ptr(), std::ptr::null_mut()) } != sys::RT_OK {
            return Err(self.error()).context("rendering");
        }
        Ok(sum)
    }

    /// The deterministic parity pass: (object id or u32::MAX, t, normal) per pixel.
    pub fn first_hit(&mut self, width: u32, height: u32) -> Result<(Vec<u32>, Vec<f64>, Vec<f64>)> {
        let n = width as usize * height as usize;
        let (mut id, mut t, mut normal) = (vec![0u32; n], vec![0f64; n], vec![0f64; 3 * n]);
        let rc = unsafe {
            sys::rt_first_hit(self.ctx, width, height, std::ptr::null(), id.as_mut_ptr(), t.as_mut_ptr(), normal.as_mut_ptr())
        };
        if rc != sys::RT_OK {
            return Err(self.error()).context("first-hit pass");
        }
        Ok((id, t, normal))
    }
}

impl Drop for Gpu {
    fn drop(&mut self) {
        unsafe { sys::rt_destroy(self.ctx) }
    }
}
