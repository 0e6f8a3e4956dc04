// Append to src/raytracing/camera.rs (the fields of camera.rs:4-10 are private to that module).
impl Camera {
    pub fn position(&self) -> Vec3 {
        self.position
    }
    pub fn forward(&self) -> Vec3 {
        self.forward
    }
    pub fn up(&self) -> Vec3 {
        self.up
    }
    pub fn right(&self) -> Vec3 {
        self.right
    }
    pub fn zoom(&self) -> f64 {
        self.zoom
    }
}
