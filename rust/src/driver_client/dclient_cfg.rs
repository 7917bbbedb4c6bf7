//! Card description.  The reference's `DriverConfig` is a table of AXI base addresses per FPGA card
//! (`/root/reference/src/driver_client/dclient_cfg.rs`); a GPU has none, so only the card type survives.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum CardType {
    /// kept so that callers written for the FPGA card still compile; treated like `MI355X`
    C1100,
    MI355X,
}

#[derive(Debug, Clone, Copy)]
pub struct DriverConfig {
    pub card: CardType,
}

impl DriverConfig {
    /// `DriverConfig::driver_client_cfg(CardType::C1100)` in the reference's tests.
    pub fn driver_client_cfg(card: CardType) -> Self {
        DriverConfig { card }
    }
}
