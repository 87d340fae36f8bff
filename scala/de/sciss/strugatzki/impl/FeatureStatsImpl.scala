/*
 *  FeatureStatsImpl.scala -- replacement of the reference's class of this name (per-channel 1st / 99th percentiles
 *  over the given feature files, histograms on the GPU).
 */
package de.sciss.strugatzki
package impl

import java.io.File

import de.sciss.processor.impl.ProcessorImpl

import scala.collection.immutable.{IndexedSeq => Vec}
import scala.concurrent.blocking

private[strugatzki] final class FeatureStatsImpl(val config: Vec[File])
  extends FeatureStats with ProcessorImpl[FeatureStats.Product, FeatureStats] {

  protected def body(): FeatureStats.Product = blocking {
    Native.withContexts(Native.devices.take(1)) { ctx =>
      val flat = Native.run(this, p => { progress = p }, () => checkAborted()) { obs =>
        Native.runFeatureStats(ctx(0), config.map(_.getPath)(collection.breakOut), obs)
      }
      Vector.tabulate(flat.length / 2)(ch => (flat(2 * ch), flat(2 * ch + 1)))
    }
  }
}
