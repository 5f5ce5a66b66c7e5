//! The flatten layer: what the reference's loader records, next to every `Box<dyn Hittable>` it
//! pushes (src/scene/loader.rs:95-189), so that `Scene` can cross the C ABI as an `rt_scene`.
//!
//! The reference keeps `BvhNode`, `Camera` and the trait objects private (src/hittable/bvh.rs:13-26,
//! src/camera.rs:6-15), so nothing can be read back out of a realised scene; the loader is the one
//! place that still sees every primitive, wrapper, material and texture in order.  `SceneFlattener`
//! is driven from there: one call per match arm of `realize_hittable`, `realize_material` and
//! `realize_texture`.  `raytracer_amd/host/loader.cpp` is the same bookkeeping in C++ and
//! `tests/test_loader_pin.py` holds it to a second reading of the loader.
//!
//! Source only: written without a Rust toolchain at hand.
use mi355rt_sys as sys;

/// Owns the arrays an `rt_scene` points into.
#[derive(Default)]
pub struct FlatScene {
    pub objects: Vec<sys::rt_object>,
    pub prims: Vec<sys::rt_prim>,
    pub ops: Vec<sys::rt_op>,
    pub materials: Vec<sys::rt_material>,
    pub textures: Vec<sys::rt_texture>,
    pub perlin: Vec<sys::rt_perlin>,
    pub images: Vec<sys::rt_image>,
    pub texels: Vec<u8>,
    pub camera: sys::rt_camera,
    pub background: [f64; 3],
    pub image: [u32; 4], // width, height, samples_per_pixel, max_depth (src/config.rs:11-17)
}

impl FlatScene {
    /// The view the C ABI takes; valid while `self` is not moved or changed.
    pub fn view(&self) -> sys::rt_scene {
        sys::rt_scene {
            camera: self.camera,
            background: self.background,
            width: self.image[0],
            height: self.image[1],
            samples_per_pixel: self.image[2],
            max_depth: self.image[3],
            n_objects: self.objects.len() as u32,
            n_prims: self.prims.len() as u32,
            n_ops: self.ops.len() as u32,
            n_materials: self.materials.len() as u32,
            n_textures: self.textures.len() as u32,
            n_perlin: self.perlin.len() as u32,
            n_images: self.images.len() as u32,
            reserved: 0,
            n_texel_bytes: self.texels.len() as u64,
            objects: self.objects.as_ptr(),
            prims: self.prims.as_ptr(),
            ops: self.ops.as_ptr(),
            materials: self.materials.as_ptr(),
            textures: self.textures.as_ptr(),
            perlin: self.perlin.as_ptr(),
            images: self.images.as_ptr(),
            texels: self.texels.as_ptr(),
        }
    }
}

/// One realised hittable before its wrappers are known: the prims it consists of and the
/// wrappers seen so far, innermost last.  Mirrors one entry of the loader's `HittableAccum`.
#[derive(Clone, Default)]
pub struct Pending {
    prims: Vec<sys::rt_prim>,
    cuboid: Option<([f64; 3], [f64; 3])>,
    /// wrappers outside the medium (or all of them when there is no medium), outermost first
    outer: Vec<sys::rt_op>,
    /// `Some((neg_inv_density, isotropic material))` once a ConstantMedium wrapped this object;
    /// the wrappers recorded before that moment are `inner`
    medium: Option<(f64, u32)>,
    inner: Vec<sys::rt_op>,
}

#[derive(Default)]
pub struct SceneFlattener {
    pub flat: FlatScene,
}

impl SceneFlattener {
    // ---- textures and materials: indices stand in for Box<dyn Texture> / Arc<dyn Material> ----

    /// `TextureDesc::Solid(r, g, b)` (loader.rs:358).
    pub fn solid(&mut self, c: [f64; 3]) -> u32 {
        self.push_texture(sys::rt_texture { kind: sys::RT_TEX_SOLID, color: c, ..Default::default() })
    }
    /// `TextureDesc::Checkerboard(even, odd)` (loader.rs:359-362).
    pub fn checker(&mut self, even: u32, odd: u32) -> u32 {
        self.push_texture(sys::rt_texture { kind: sys::RT_TEX_CHECKER, a: even, b: odd, ..Default::default() })
    }
    /// `TextureDesc::Perlin` (loader.rs:363): the tables `Perlin::new` drew (perlin.rs:17-36), scale 4.
    pub fn perlin(&mut self, tables: sys::rt_perlin) -> u32 {
        self.flat.perlin.push(tables);
        let a = self.flat.perlin.len() as u32 - 1;
        self.push_texture(sys::rt_texture { kind: sys::RT_TEX_PERLIN, a, scale: 4.0, ..Default::default() })
    }
    /// `TextureDesc::Image(path)` (loader.rs:364-375): the decoded RGB8 rows, top row first.
    pub fn image(&mut self, width: u32, height: u32, rgb8: &[u8]) -> u32 {
        let offset = self.flat.texels.len() as u64;
        self.flat.texels.extend_from_slice(rgb8);
        self.flat.images.push(sys::rt_image { width, height, offset });
        let a = self.flat.images.len() as u32 - 1;
        self.push_texture(sys::rt_texture { kind: sys::RT_TEX_IMAGE, a, ..Default::default() })
    }
    fn push_texture(&mut self, t: sys::rt_texture) -> u32 {
        self.flat.textures.push(t);
        self.flat.textures.len() as u32 - 1
    }

    /// `Lambertian { albedo }` / `DiffuseLight { texture }` (loader.rs:323-325, :338-340).
    pub fn textured(&mut self, kind: u32, texture: u32) -> u32 {
        self.push_material(sys::rt_material { kind, texture, ..Default::default() })
    }
    /// `Metal { albedo, fuzziness }` (loader.rs:334-337).
    pub fn metal(&mut self, albedo: [f64; 3], fuzziness: f64) -> u32 {
        self.push_material(sys::rt_material { kind: sys::RT_MAT_METAL, albedo, param: fuzziness, ..Default::default() })
    }
    /// `Dielectric { index_of_refraction }` (loader.rs:326-333).
    pub fn dielectric(&mut self, index_of_refraction: f64) -> u32 {
        self.push_material(sys::rt_material { kind: sys::RT_MAT_DIELECTRIC, param: index_of_refraction, ..Default::default() })
    }
    fn push_material(&mut self, m: sys::rt_material) -> u32 {
        self.flat.materials.push(m);
        self.flat.materials.len() as u32 - 1
    }

    // ---- hittables: one constructor per arm of realize_hittable ----

    /// `hittable::Sphere` (loader.rs:101-109).
    pub fn sphere(center: [f64; 3], radius: f64, material: u32) -> Pending {
        Pending {
            prims: vec![sys::rt_prim { kind: sys::RT_PRIM_SPHERE, center, radius, material, ..Default::default() }],
            ..Default::default()
        }
    }
    /// `hittable::MovingSphere` (loader.rs:111-121).
    pub fn moving_sphere(c0: [f64; 3], c1: [f64; 3], t0: f64, t1: f64, radius: f64, material: u32) -> Pending {
        Pending {
            prims: vec![sys::rt_prim {
                kind: sys::RT_PRIM_MOVING_SPHERE, center: c0, center1: c1, time0: t0, time1: t1, radius, material,
                ..Default::default()
            }],
            ..Default::default()
        }
    }
    /// `AxisAlignedRect` (loader.rs:123-135); axis 0 = X, 1 = Y, 2 = Z.
    pub fn rect(center: [f64; 3], width: f64, height: f64, axis: u32, material: u32) -> Pending {
        Pending {
            prims: vec![sys::rt_prim { kind: sys::RT_PRIM_RECT, axis, center, width, height, material, ..Default::default() }],
            ..Default::default()
        }
    }
    /// `Cuboid::new` (loader.rs:137-145, cuboid.rs:18-37): the six sides in the order
    /// `iproduct!([X, Y, Z], [-1.0, 1.0])`, `width = size[axis.next()]`, `height = size[axis.prev()]`.
    /// `half` is `size / 2.0` and `side_centers[k]` is `center + half.project(axis) * mult`, both
    /// computed by the reference's own `Vec3` so that every bit agrees.
    pub fn cuboid(center: [f64; 3], size: [f64; 3], half: [f64; 3], side_centers: [[f64; 3]; 6], material: u32) -> Pending {
        let mut prims = Vec::with_capacity(6);
        for axis in 0..3u32 {
            for s in 0..2usize {
                prims.push(sys::rt_prim {
                    kind: sys::RT_PRIM_RECT,
                    axis,
                    center: side_centers[axis as usize * 2 + s],
                    width: size[((axis + 1) % 3) as usize],
                    height: size[((axis + 2) % 3) as usize],
                    material,
                    ..Default::default()
                });
            }
        }
        Pending { prims, cuboid: Some((center, half)), ..Default::default() }
    }

    /// `Translate { offset, hittable }` over every object of `inner` (loader.rs:151-160).
    pub fn translate(offset: [f64; 3], inner: Vec<Pending>) -> Vec<Pending> {
        Self::wrap(sys::rt_op { kind: sys::RT_OP_TRANSLATE, reserved: 0, a: offset }, inner)
    }
    /// `RotateY::new(h, theta)` (loader.rs:163-173): `(sin, cos) = theta.sin_cos()` (instance.rs:38).
    pub fn rotate_y(sin_theta: f64, cos_theta: f64, inner: Vec<Pending>) -> Vec<Pending> {
        Self::wrap(sys::rt_op { kind: sys::RT_OP_ROTATE_Y, reserved: 0, a: [sin_theta, cos_theta, 0.0] }, inner)
    }
    fn wrap(op: sys::rt_op, mut inner: Vec<Pending>) -> Vec<Pending> {
        for p in &mut inner {
            p.outer.insert(0, op); // the newest wrapper is the outermost
        }
        inner
    }
    /// `ConstantMedium::new(h, density, texture)` over every object of `inner` (loader.rs:175-186):
    /// `neg_inv_density = -density.recip()`, phase function `Isotropic { albedo: texture }`.
    pub fn constant_medium(&mut self, density: f64, texture: u32, mut inner: Vec<Pending>) -> anyhow::Result<Vec<Pending>> {
        let iso = self.push_material(sys::rt_material { kind: sys::RT_MAT_ISOTROPIC, texture, ..Default::default() });
        for p in &mut inner {
            anyhow::ensure!(p.medium.is_none(), "a ConstantMedium whose boundary is a ConstantMedium is not representable");
            p.inner = std::mem::take(&mut p.outer);
            p.medium = Some((-density.recip(), iso));
        }
        Ok(inner)
    }

    /// The loader's `hittables.add(..)` / `add_many(..)` at top level: the object's index in
    /// `objects[]` is its id in the first-hit pass.
    pub fn add(&mut self, pending: Vec<Pending>) {
        for p in pending {
            let f = &mut self.flat;
            let mut o = sys::rt_object {
                first_prim: f.prims.len() as u32,
                prim_count: p.prims.len() as u32,
                first_op: f.ops.len() as u32,
                op_count: p.outer.len() as u32,
                ..Default::default()
            };
            f.ops.extend_from_slice(&p.outer);
            o.inner_first_op = f.ops.len() as u32;
            o.inner_op_count = p.inner.len() as u32;
            f.ops.extend_from_slice(&p.inner);
            if let Some((nid, iso)) = p.medium {
                o.is_medium = 1;
                o.neg_inv_density = nid;
                o.medium_material = iso;
            }
            if let Some((c, h)) = p.cuboid {
                o.is_cuboid = 1;
                o.cuboid_center = c;
                o.cuboid_half = h;
            }
            f.prims.extend_from_slice(&p.prims);
            f.objects.push(o);
        }
    }

    /// `Camera` after `done()` (needs `pub(crate)` getters on src/camera.rs:6-15) and the rest of `Scene`.
    pub fn finish(mut self, camera: sys::rt_camera, background: [f64; 3], image: [u32; 4]) -> FlatScene {
        self.flat.camera = camera;
        self.flat.background = background;
        self.flat.image = image;
        self.flat
    }
}
